"""The two depth-wise weight-gradient kernels side by side (MAF_DWWG_MFMA=0: the vector kernel of train_ops.hip, =1: the matrix-core kernel of dw_wgrad_mfma.hip)
over the shapes of a MAF-YOLO-n step at batch 32: microseconds per launch (HIP events, 20 launches) and the worst deviation from the framework's fp32 weight gradient
of the same fp16 tensors (in units of the gradient's max |g|).    python tools/dw_wgrad_ab.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(160, 72, 3), (80, 192, 5), (80, 192, 3), (80, 144, 5), (80, 144, 3), (80, 128, 5), (80, 128, 3), (40, 288, 7), (40, 288, 5), (40, 288, 3), (40, 192, 7), (40, 192, 5), (40, 192, 3),
          (40, 128, 7), (40, 128, 5), (40, 128, 3), (20, 576, 9), (20, 576, 7), (20, 576, 5), (20, 576, 3), (20, 288, 9), (20, 288, 7), (20, 288, 5), (20, 288, 3), (20, 192, 9), (20, 192, 7),
          (20, 192, 5), (20, 192, 3)]
if len(sys.argv) > 1:
    from maf_yolo_amd import lib
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B = int(os.environ.get("BATCH", "32"))
    for (H, C, k) in SHAPES:
        g = torch.Generator().manual_seed(H + C + k)
        x = torch.randn(B, H, H, C, generator=g).half().cuda()
        dy = torch.randn(B, H, H, C, generator=g).half().cuda()
        R = 8
        dw = torch.zeros(R, C, k * k, device="cuda")
        f = lambda: lib.check(L.maf_dw_wgrad(x.data_ptr(), C, dy.data_ptr(), C, B, H, H, C, k, lib.F16, dw.data_ptr(), R, st))
        f()
        torch.cuda.synchronize()
        got = dw.sum(0).view(C, 1, k, k)
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (C, 1, k, k), dy.permute(0, 3, 1, 2).float(), padding=k // 2, groups=C)
        err = float((got - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("%d %d %d %.1f %.2e" % (H, C, k, e0.elapsed_time(e1) / 20 * 1e3, err))
    sys.exit(0)
res = {}
modes = os.environ.get("MODES", "0 1").split()
for m in modes:
    env = dict(os.environ)
    env["MAF_DWWG_MFMA"] = m
    out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
    if out.returncode:
        print(out.stderr[-2000:])
    for line in out.stdout.strip().splitlines():
        *k, t, e = line.split()
        res.setdefault(tuple(k), {})[m] = (float(t), float(e))
print("| map | C | k | " + " | ".join("us (mode %s) | err" % m for m in modes) + " |")
print("|---|---|---|" + "---|---|" * len(modes))
tot = {m: 0.0 for m in modes}
best = 0.0
for k, v in res.items():
    print("| %sx%s | %s | %s | " % (k[0], k[0], k[1], k[2]) + " | ".join("%.1f | %.1e" % v.get(m, (-1, -1)) for m in modes) + " |")
    for m in modes:
        tot[m] += v.get(m, (0, 0))[0]
    best += min(v[m][0] for m in modes if m in v)
print("sum: " + "  ".join("%s: %.0f us" % (m, tot[m]) for m in modes) + "  best of each: %.0f us" % best)
