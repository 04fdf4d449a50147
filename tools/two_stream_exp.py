"""Experiment: forwards of consecutive batches on alternating streams with separate plans (arenas): do the latency-bound small-map kernels
of one forward fill the chip under the other's?  python tools/two_stream_exp.py"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
engine = importlib.import_module("maf-yolo_amd.engine")
dev = torch.device("cuda:0")
if os.path.exists("profiles/round1_tune.json"):
    engine.load_tune_cache("profiles/round1_tune.json")
models = []
for _ in range(2):
    m = M.Model("n"); m.load_state_dict(synth.synth_state_dict(m, "n", 0)); m = m.to(dev).eval().half(); m.autotune = True
    models.append(m)
x = synth.synth_images(32, 640, seed=1).to(dev).half()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
with torch.no_grad():
    for m in models: m(x)
torch.cuda.synchronize()
def run(n, two):
    outs = [None, None]
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            k = i % 2 if two else 0
            with torch.cuda.stream(streams[k]):
                outs[k] = models[k](x)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    run(10, two)
    print("two streams" if two else "one stream ", "%.4f ms / forward" % run(60, two), flush=True)
