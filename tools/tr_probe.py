"""What does ds_read_b64_tr_b16 deliver?  LDS half i = i; lane l of a 16-lane group reads row (l >> 2) [+ 8 * group], columns 4 * (l & 3) .. +3 of
a [row][S] image.  Expectation used by csrc/wgrad.hip: lane i, element j = image[row j of the group's 4 rows][column i]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib  # noqa: E402

lib.load()                                        # the product library (the probe library links it)
L = __import__("ctypes").CDLL(__import__("os").path.join(__import__("os").path.dirname(lib.LIB_PATH), "libmafyolo_probe.so"))   # `make -C maf-yolo_amd/csrc probe` (tools/probe.hip: not in the product)
S = 72                                    # row stride in halfs
lanes = torch.arange(64)
g, p = lanes // 16, lanes % 16
addr = ((g * 8 + (p >> 2)) * S + (p & 3) * 4) * 2
a = addr.int().cuda()
out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
L.maf_probe_tr.argtypes = [__import__("ctypes").c_void_p] * 3
lib.check(L.maf_probe_tr(torch.cuda.current_stream().cuda_stream, a.data_ptr(), out.data_ptr()))
torch.cuda.synchronize()
o = out.cpu().view(64, 4).int()
ok = True
for l in range(64):
    want = [(int(g[l]) * 8 + j) * S + int(p[l]) for j in range(4)]
    got = o[l].tolist()
    if l < 20 or got != want:
        print("lane %2d got %s (row, col) %s   want %s" % (l, got, [(v // S, v % S) for v in got], want))
    ok &= got == want
print("MATCH" if ok else "MISMATCH")
