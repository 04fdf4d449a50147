#!/usr/bin/env python3
"""Fixed cost vs streaming rate of the BatchNorm(train) kernels (csrc/bn_act.hip) over the tensor shapes of a MAF-YOLO-n step at batch 32:
per shape the statistics pass alone, the forward call (statistics + apply) and the backward call (reduction + apply), HIP events over N back-to-back calls.

    python tools/bn_bench.py [N]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maf_yolo_amd as M                    # noqa: E402
from maf_yolo_amd import lib, train_ops     # noqa: E402

SHAPES = [(160, 48), (160, 72), (80, 96), (80, 144), (80, 128), (80, 192), (40, 192), (40, 288), (40, 128), (20, 384), (20, 576), (20, 192), (20, 96)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    L = lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    print("%-14s %9s | %8s %8s | %8s %8s | %8s %8s" % ("shape", "MB", "stats us", "TB/s", "fwd us", "TB/s", "bwd us", "TB/s"))
    rows = []
    for hw, c in SHAPES:
        B = 32
        x = torch.randn(B, c, hw, hw, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        dz = torch.randn_like(x)
        bn = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.03).to(dev).train()
        Mpix = B * hw * hw
        mb = x.numel() * 2 / 1e6
        part = torch.zeros(2 * 16 * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev)
        t = lib.Timer()

        def timed(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t.start(st)
            for _ in range(n):
                fn()
            t.stop(st)
            return 1e3 * t.elapsed_ms() / n

        ph = [0]

        def stats():
            lib.check(L.maf_bn_stats(x.data_ptr(), c, Mpix, c, lib.F16, part.data_ptr(), 16, ph[0], st))
        us_s = timed(stats)
        part.zero_()
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        stat = torch.empty(2, c, dtype=torch.float32, device=dev)
        dgb = torch.empty(2, c, dtype=torch.float32, device=dev)
        g, b_ = bn.weight.detach(), bn.bias.detach()
        sp = stat.data_ptr()

        def fwd():                                               # straight through the C-ABI: the host must not be the bound of a 15 us call
            ph[0] ^= 1
            lib.check(L.maf_bn_forward_ex(x.data_ptr(), c, Mpix, c, lib.F16, g.data_ptr(), b_.data_ptr(), 1e-3, 0.03, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                          bn.num_batches_tracked.data_ptr(), lib.ACT_SILU, y.data_ptr(), c, sp, sp + 4 * c, part.data_ptr(), 16, ph[0], None, 0, 0, st))

        def bwd():
            ph[0] ^= 1
            lib.check(L.maf_bn_backward_acc(x.data_ptr(), c, dz.data_ptr(), c, Mpix, c, lib.F16, g.data_ptr(), b_.data_ptr(), sp, sp + 4 * c, lib.ACT_SILU, dx.data_ptr(), c,
                                            dgb.data_ptr(), dgb.data_ptr() + 4 * c, part.data_ptr(), 16, ph[0], None, 0, None, 0, 0, st))
        us_f = timed(fwd)
        us_b = timed(bwd)
        rows.append((hw, c, mb, us_s, us_f, us_b))
        print("%3dx%-3d x %-4d %9.1f | %8.1f %8.2f | %8.1f %8.2f | %8.1f %8.2f" % (hw, hw, c, mb, us_s, mb / us_s, us_f, 3 * mb / us_f, us_b, 5 * mb / us_b))
    # least-squares fit  t = a + bytes / bw  over the shapes
    import numpy as np
    for name, col, passes in (("stats", 3, 1), ("forward", 4, 3), ("backward", 5, 5)):
        A = np.array([[1.0, passes * r[2]] for r in rows])
        b = np.array([r[col] for r in rows])
        (a0, k), *_ = np.linalg.lstsq(A, b, rcond=None)
        print("%-8s: t = %.1f us + bytes / %.2f TB/s" % (name, a0, 1.0 / k))


if __name__ == "__main__":
    main()
