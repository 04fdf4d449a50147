#!/usr/bin/env python3
"""Do kernels launched with hipExtAnyOrderLaunch (no AQL barrier bit) overlap their predecessors in the same stream on this device?"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib   # noqa: E402

L = lib.load()
L.maf_probe_anyorder.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_float)]
torch.zeros(1, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for blocks in (32, 128, 512):
    for n in (1, 4, 8):
        row = []
        for flags in (0, 1):
            ms = C.c_float()
            best = 1e9
            for _ in range(5):
                lib.check(L.maf_probe_anyorder(st, n, blocks, 100000, flags, C.byref(ms)))
                best = min(best, ms.value)
            row.append(best * 1e3)
        print("blocks %4d  n %d:  in order %8.1f us   any-order %8.1f us" % (blocks, n, row[0], row[1]))
