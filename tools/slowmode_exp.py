"""Which part of the two-in-flight loop has the slow mode?  python tools/slowmode_exp.py  (run several times)"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
engine = importlib.import_module("maf-yolo_amd.engine")
dev = torch.device("cuda:0")
engine.load_tune_cache("profiles/round1_tune.json")
m = M.Model("n"); m.load_state_dict(synth.synth_state_dict(m, "n", 0)); m = m.to(dev).eval().half(); m.autotune = True
x = synth.synth_images(32, 640, seed=1).to(dev).half()
B.calibrate_cls_bias(m, x, 2000, M, torch)
cs = M.concurrent_streams(dev, 3)
def loop(n, nms, streams):
    pending = []
    with torch.no_grad():
        for i in range(n):
            k = i % 2
            with torch.cuda.stream(streams[k]):
                p = m(x, slot=k)[0]
                if nms: pending.append(M.non_max_suppression_async(p, 0.03, 0.65, multi_label=True, side=cs[2]))
            if len(pending) > 2: pending.pop(0).result()
    for h in pending: h.result()
    torch.cuda.synchronize()
def rate(nms, n=50):
    loop(10, nms, cs[:2]); t0 = time.perf_counter(); loop(n, nms, cs[:2]); return (time.perf_counter() - t0) / n * 1e3
r = [rate(True), rate(False), rate(True), rate(False)]
print("with NMS %.3f  forward only %.3f  with NMS %.3f  forward only %.3f ms/step" % tuple(r), flush=True)
