#!/usr/bin/env python3
"""Where the time of the fused DepthBottleneckUni kernel goes: runs the profiling build (make -C maf-yolo_amd/csrc prof) on the bench's
bottleneck shapes with one piece knocked out at a time (op->aux[2] bit mask; results are wrong, the time difference is the piece's cost).

  MAF_HIP_LIB=maf-yolo_amd/libmafyolo_prof.so python tools/bn_profile.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MAF_HIP_LIB", os.path.join(ROOT, "maf-yolo_amd", "libmafyolo_prof.so"))
from maf_yolo_amd import lib, pack          # noqa: E402

dev = "cuda:0"
KO = [("full kernel", 0), ("- activation re-reads (blocks > 0)", 1), ("- SiLU in phase A", 2), ("- phase B (depth-wise MFMA)", 4), ("- SiLU in phase C", 8),
      ("- T1 stores", 16), ("- operand DMA (blocks > 0)", 32), ("- both SiLUs", 2 | 8), ("- re-reads, SiLUs", 1 | 2 | 8), ("- re-reads, SiLUs, B", 1 | 2 | 4 | 8), ("- everything so far", 63),
      ("- output stores", 64), ("- barriers in the loop", 128), ("- phase A", 256), ("- second 1x1 MFMAs", 512), ("- all + output stores", 63 | 64), ("- all + stores + barriers", 63 | 64 | 128),
      ("- all + stores + barriers + A", 63 | 64 | 128 | 256), ("- all of the above", 1023)]
for (c, k, hw, B) in ((64, 5, 80, 32), (48, 5, 80, 32), (24, 3, 160, 32)):
    g = torch.Generator().manual_seed(1)
    mid = 3 * c
    w1 = torch.randn(mid, c, 1, 1, generator=g) / c ** 0.5; b1 = torch.randn(mid, generator=g) * 0.3
    wd = torch.randn(mid, 1, k, k, generator=g) / k; bd = torch.randn(mid, generator=g) * 0.3
    w2 = torch.randn(c, mid, 1, 1, generator=g) / mid ** 0.5; b2 = torch.randn(c, generator=g) * 0.3
    rec, b2p, nmb, ct2 = pack.pack_bottleneck(w1, b1, wd, bd, w2, b2)
    recd, b2d = rec.to(dev), b2p.to(dev)
    stride = 3 * c
    buf = torch.randn(B, hw, hw, stride, generator=g).half().to(dev)
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = lib.OP_BOTTLENECK, lib.F16, lib.F16, lib.ACT_SILU
    op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc = B, hw, hw, c, c, k, 1
    op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff = buf.data_ptr(), c, stride, c
    op.out, op.out_stride, op.out_coff = buf.data_ptr(), stride, 2 * c
    op.tile_p, op.tile_c, op.tile_k = 16, 16, nmb
    op.w, op.bias = recd.data_ptr(), b2d.data_ptr()
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    t = lib.Timer()
    print("c=%d k=%d %dx%d B=%d (%d mid blocks)" % (c, k, hw, hw, B, nmb))
    for name, ko in KO:
        op.aux[2] = ko
        ts = []
        for i in range(8):
            t.start(st)
            lib.check(L.maf_op_launch(C.byref(op), st))
            t.stop(st)
            ts.append(t.elapsed_ms() * 1e3)
        print("   %-36s %7.1f us" % (name, min(ts[2:])))
