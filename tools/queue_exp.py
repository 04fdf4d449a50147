"""Does the two-in-flight rate depend on which hardware queues the two streams land on?  python tools/queue_exp.py"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
engine = importlib.import_module("maf-yolo_amd.engine")
dev = torch.device("cuda:0")
if os.path.exists("profiles/round1_tune.json"):
    engine.load_tune_cache("profiles/round1_tune.json")
m = M.Model("n"); m.load_state_dict(synth.synth_state_dict(m, "n", 0)); m = m.to(dev).eval().half(); m.autotune = True
x = synth.synth_images(32, 640, seed=1).to(dev).half()
with torch.no_grad():
    for k in range(2): m(x, slot=k)
torch.cuda.synchronize()
keep = []
def rate(streams, n=60):
    def loop(n):
        with torch.no_grad():
            for i in range(n):
                with torch.cuda.stream(streams[i % 2]):
                    m(x, slot=i % 2)
    loop(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for trial in range(8):
    s = [torch.cuda.Stream(dev) for _ in range(2)]
    keep += s
    print("streams created so far %2d: %.4f ms / forward" % (len(keep), rate(s)), flush=True)
    keep.append(torch.cuda.Stream(dev)) if trial % 2 else None
for trial in range(4):
    s = M.concurrent_streams(dev, 2)
    print("concurrent_streams pair %d: %.4f ms / forward" % (trial, rate(s)), flush=True)
    keep.append(torch.cuda.Stream(dev))
