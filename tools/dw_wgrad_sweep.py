"""Sweep of the depth-wise weight-gradient launch choices (MAF_DWWG = tile20,gmax,maxthreads,wgcap): python tools/dw_wgrad_sweep.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(160, 72, 3), (80, 192, 3), (80, 192, 5), (80, 128, 5), (40, 288, 7), (40, 192, 7), (20, 576, 9), (20, 288, 9), (20, 192, 5), (40, 128, 3)]
if os.environ.get("SMALL"):                   # the small maps only (round 5: whole-plane tiles)
    SHAPES = [(20, 576, 9), (20, 576, 7), (20, 576, 5), (20, 576, 3), (20, 288, 9), (20, 288, 7), (20, 192, 9), (20, 192, 7), (40, 288, 7), (40, 288, 5), (40, 192, 7), (40, 192, 5), (40, 128, 7), (40, 128, 3)]
if len(sys.argv) > 1:
    from maf_yolo_amd import lib
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for (H, C, k) in SHAPES:
        x = torch.randn(32, H, H, C, device="cuda").half()
        dy = torch.randn(32, H, H, C, device="cuda").half()
        R = int(os.environ.get("REPS", "32"))
        dw = torch.zeros(R, C, k * k, device="cuda")
        f = lambda: lib.check(L.maf_dw_wgrad(x.data_ptr(), C, dy.data_ptr(), C, 32, H, H, C, k, lib.F16, dw.data_ptr(), R, st))
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("%d %d %d %.1f" % (H, C, k, e0.elapsed_time(e1) / 20 * 1e3))
    sys.exit(0)
res = {}
cfgs = os.environ.get("CFGS", "0,8,256,1024 0,4,256,1024 0,2,256,1024 0,4,384,1024 0,4,256,2048 0,4,256,4096 0,2,256,2048 0,4,128,1024 0,3,256,1024 0,4,320,2048").split()
for cfg in cfgs:
    env = dict(os.environ)
    env["MAF_DWWG"] = cfg
    out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        *k, t = line.split()
        res.setdefault(tuple(k), {})[cfg] = float(t)
print("configs:", cfgs)
tot = {c: 0.0 for c in cfgs}
for k, v in res.items():
    print("%sx%s C=%s k%s: " % (k[0], k[0], k[1], k[2]) + "  ".join("%.0f" % v.get(c, -1) for c in cfgs))
    for c in cfgs:
        tot[c] += v.get(c, 0)
print("sum: " + "  ".join("%.0f" % tot[c] for c in cfgs))
