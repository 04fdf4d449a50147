"""Host time per serving-loop step (is the two-in-flight loop host-bound?): python tools/host_cost.py"""
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
for B in (32, 1):
    x = synth.synth_images(B, 640, seed=1).to(dev).half()
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    def loop(n, timing):
        pending = []
        t_f = t_n = t_r = 0.0
        for i in range(n):
            k = i % 2
            with torch.cuda.stream(streams[k]), torch.no_grad():
                t0 = time.perf_counter(); p = m(x, slot=k)[0]; t1 = time.perf_counter()
                pending.append(M.non_max_suppression_async(p, 0.03, 0.65, multi_label=True)); t2 = time.perf_counter()
            if len(pending) > 2:
                pending.pop(0).result()
            t3 = time.perf_counter()
            t_f += t1 - t0; t_n += t2 - t1; t_r += t3 - t2
        for h in pending: h.result()
        return t_f / n * 1e3, t_n / n * 1e3, t_r / n * 1e3
    loop(10, False); torch.cuda.synchronize()
    t0 = time.perf_counter(); f, nn, r = loop(100, True); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100 * 1e3
    print("B=%d: wall %.3f ms/step; host inside forward call %.3f, nms_async call %.3f, result() %.3f ms" % (B, wall, f, nn, r), flush=True)
