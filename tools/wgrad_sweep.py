"""Pixel-chunk count (grid x) sweep of the conv weight-gradient kernel on shapes of a MAF-YOLO-n step: python tools/wgrad_sweep.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(80, 288, 128, 1), (80, 256, 128, 1), (40, 192, 192, 3), (80, 128, 128, 3), (20, 768, 384, 1), (40, 576, 128, 1), (160, 48, 64, 3), (80, 192, 64, 1),
          (20, 288, 192, 1), (160, 72, 48, 1), (40, 128, 128, 3)]
if len(sys.argv) > 1:
    from maf_yolo_amd import lib
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for (H, cin, cout, k) in SHAPES:
        B = 32
        s = 2 if k == 3 else 1
        Ho = H // s
        x = torch.randn(B, H, H, cin, device="cuda").half()
        dy = torch.randn(B, Ho, Ho, cout, device="cuda").half()
        dw = torch.zeros(k, k, cout, cin, device="cuda")
        f = lambda: lib.check(L.maf_conv_wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, B, Ho, Ho, H, H, cin, cout, k, s, lib.F16, dw.data_ptr(), st))
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("%d %d %d %d %.1f" % (H, cin, cout, k, e0.elapsed_time(e1) / 20 * 1e3))
    sys.exit(0)
res = {}
for gx in ["default", 2, 4, 8, 16, 32, 64, 128, 256]:
    env = dict(os.environ)
    if gx != "default":
        env["MAF_WGRAD_GX"] = str(gx)
    out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        *k, t = line.split()
        res.setdefault(tuple(k), {})[gx] = float(t)
for k, v in res.items():
    print("in %sx%s %s->%s k%s: " % (k[0], k[0], k[1], k[2], k[3]) + "  ".join("%s:%.0f" % (g, t) for g, t in v.items()))
