import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maf_yolo_amd as M
from maf_yolo_amd import synth, engine
import bench
dev = torch.device("cuda:0")
engine.load_tune_cache("profiles/round5_tune.json")
model = M.Model("n"); model.load_state_dict(synth.synth_state_dict(model, "n", 0)); model = model.to(dev).eval(); model.autotune = True
S = 3
xs = [synth.synth_images(32, 640, seed=1 + 7 * k).to(dev).half() for k in range(S)]
bench.calibrate_cls_bias(model, xs[0], 2000, M, torch)
cs = M.concurrent_streams(dev, S + 1)
def loop(n, nms):
    pend = []
    for i in range(n):
        k = i % S
        with torch.cuda.stream(cs[k]), torch.no_grad():
            p = model(xs[k], slot=k)[0]
            if nms:
                pend.append(M.non_max_suppression_async(p, 0.03, 0.65, multi_label=True, side=cs[S]))
        if len(pend) > S: pend.pop(0).result()
    for h in pend: h.result()
for nms in (True, False, True, False):
    loop(60, nms); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(300, nms); torch.cuda.synchronize()
    print("nms", nms, "ms/step %.4f" % ((time.perf_counter() - t0) / 300 * 1e3))
