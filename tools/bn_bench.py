"""BatchNorm(train)+activation kernels (csrc/bn_act.hip) on the activation shapes of a MAF-YOLO-n step at batch 32, called through the C-ABI
back to back (no autograd): GPU time per call (two launches) and achieved HBM rate (forward = 3 passes of the tensor, backward = 5).
    python tools/bn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib  # noqa: E402

L = lib.load()
BS = int(os.environ.get("BS", "32"))
R = 16
st = torch.cuda.current_stream().cuda_stream
ACT = {None: lib.ACT_NONE, "relu": lib.ACT_RELU, "silu": lib.ACT_SILU}
for (H, c, act) in [(320, 24, None), (160, 72, "silu"), (160, 48, None), (80, 192, "silu"), (80, 144, None), (80, 128, "silu"), (320, 16, "silu"), (160, 32, "silu"), (160, 48, None), (160, 24, "relu"), (80, 64, "silu"), (80, 192, None), (80, 96, "silu"),
                    (40, 128, "silu"), (40, 384, None), (20, 256, "silu"), (20, 768, None)]:
    M = BS * H * H
    x = torch.randn(M, c, device="cuda").half()
    dz = torch.randn(M, c, device="cuda").half()
    y = torch.empty_like(x)
    g, b = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    stat, dgb = torch.empty(2, c, device="cuda"), torch.empty(2, c, device="cuda")
    part = torch.zeros(2 * R * 2 * (-(-c // 256) * 256), device="cuda")
    ph = [0]

    def fwd():
        ph[0] ^= 1
        lib.check(L.maf_bn_forward(x.data_ptr(), c, M, c, lib.F16, g.data_ptr(), b.data_ptr(), 1e-3, 0.03, rm.data_ptr(), rv.data_ptr(), None, ACT[act],
                                   y.data_ptr(), c, stat[0].data_ptr(), stat[1].data_ptr(), part.data_ptr(), R, ph[0], None, 0, st))

    def bwd():
        ph[0] ^= 1
        lib.check(L.maf_bn_backward(x.data_ptr(), c, dz.data_ptr(), c, M, c, lib.F16, g.data_ptr(), b.data_ptr(), stat[0].data_ptr(), stat[1].data_ptr(),
                                    ACT[act], y.data_ptr(), c, dgb[0].data_ptr(), dgb[1].data_ptr(), part.data_ptr(), R, ph[0], None, 0, None, 0, st))

    res = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e-3)
    nbytes = x.numel() * 2
    print("bn_act %dx%dx%dx%d %-5s fwd %7.1f us (%.2f TB/s)   bwd %7.1f us (%.2f TB/s)" % (BS, H, H, c, act, res[0] * 1e6, 3 * nbytes / res[0] / 1e12,
                                                                                       res[1] * 1e6, 5 * nbytes / res[1] / 1e12))
