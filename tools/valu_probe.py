#!/usr/bin/env python3
"""Cycles per wave64 vector instruction on this device (csrc/probe.hip:maf_probe_valu): v_fma_f32, v_exp_f32, v_rcp_f32 and the SiLU sequence."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib   # noqa: E402

lib.load()                                        # the product library (the probe library links it)
L = __import__("ctypes").CDLL(__import__("os").path.join(__import__("os").path.dirname(lib.LIB_PATH), "libmafyolo_probe.so"))   # `make -C maf-yolo_amd/csrc probe` (tools/probe.hip: not in the product)
L.maf_probe_valu_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_float)]
torch.zeros(1, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
iters = 4096
for wg in (1, 2, 4, 8):
    row = []
    for kind, name in ((0, "fma"), (1, "exp+mul"), (2, "rcp+add"), (3, "silu(4 ops + add)"), (4, "fma_mix"), (5, "dot2(+2 ops)"), (6, "pk_fma_f16"), (7, "pk_fma_f32"), (8, "dot2c"), (9, "fma_mix sgpr"), (10, "fma sgpr"), (11, "dot2c sgpr")):
        c, ms = C.c_longlong(), C.c_float()
        lib.check(L.maf_probe_valu_ms(st, kind, iters, wg, C.byref(c), C.byref(ms)))
        # the counter (s_memtime) against the wall clock of the launch: ns per wave-instruction of ONE SIMD = launch time / (instructions per wave x waves per SIMD)
        row.append("%s %.2f (%.2f ns/SIMD-instr)" % (name, c.value / (iters * 8.0), ms.value * 1e6 / (iters * 8.0 * wg)))
    print("waves/SIMD %d: cycles per (lane-parallel) step:  %s" % (wg, "   ".join(row)))
