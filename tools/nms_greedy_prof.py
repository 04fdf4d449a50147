#!/usr/bin/env python3
"""Where the kept-list NMS scan spends its cycles (image 0 of the bench's batch): an instrumented build of csrc/nms.hip (-DMAF_NMS_PROF: s_memtime stamps around the
decide step, the tests and the two barriers of every block), `make -C maf-yolo_amd/csrc var VAR=nmsprof VARSRC=nms.hip VARFLAGS="-DMAF_NMS_PROF -ffp-contract=off"`,
loaded through MAF_HIP_LIB.  Never the product library."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MAF_HIP_LIB"] = os.path.join(ROOT, "maf-yolo_amd", "libmafyolo_nmsprof.so")
sys.path.insert(0, ROOT)
import torch
import maf_yolo_amd as M
from maf_yolo_amd import synth, lib, nms as nms_mod
import bench
dev = torch.device("cuda:0")
model = M.Model("n"); model.load_state_dict(synth.synth_state_dict(model, "n", 0)); model = model.to(dev).eval()
x = synth.synth_images(32, 640, seed=1).to(dev).half()
bench.calibrate_cls_bias(model, x, 2000, M, torch)
with torch.no_grad():
    pred = model(x)[0]
nms_mod.MATRIX_PATH = False
for _ in range(3):
    rows, idx, cnt = nms_mod.nms_raw(pred, 0.03, 0.65, multi_label=True)
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
L = lib.load()
L.maf_nms_debug.argtypes = [C.POINTER(C.c_uint64)]
lib.check(L.maf_nms_debug(buf))
v = list(buf)
mhz = 100.0          # s_memtime / readcyclecounter ticks: 100 MHz constant clock on gfx950
print("image 0: blocks %d, kept %d" % (v[7] >> 32, v[7] & 0xffffffff))
print("wave 0 (decider): decide %.1f us, barrier A %.1f us, stage B tests %.1f us, kernel body %.1f us" % (v[0] / mhz, v[1] / mhz, v[2] / mhz, v[3] / mhz))
print("wave 1 (tester) : tests  %.1f us, barrier A %.1f us, barrier B %.1f us" % (v[4] / mhz, v[5] / mhz, v[6] / mhz))
