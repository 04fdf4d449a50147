"""maf_dw_wgrad31 (the 3x3 [+ 3x3] + 1x1 depth-wise weight gradients of a DilatedReparamBlock in one launch) against the separate maf_dw_wgrad launches on the shapes of
a MAF-YOLO-n step at batch 32, over MAF_DWWG31 = "channel groups per block,max threads,workgroup cap,atomics budget".      python tools/dw_wgrad31_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib          # noqa: E402

SHAPES = [(160, 72, True), (80, 192, False), (80, 144, False), (80, 128, False)]


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B = 32
    for (H, c, two) in SHAPES:
        g = torch.Generator().manual_seed(H + c)
        x, da, db, d1 = [torch.randn(B, H, H, c, generator=g).half().cuda() for _ in range(4)]
        wa, wb, w1 = torch.zeros(c, 9, device="cuda"), torch.zeros(c, 9, device="cuda"), torch.zeros(c, device="cuda")
        sep = lambda d, w, k: lib.check(L.maf_dw_wgrad(x.data_ptr(), c, d.data_ptr(), c, B, H, H, c, k, lib.F16, w.data_ptr(), 1, st))
        t_sep = timed(lambda: sep(da, wa, 3)) + timed(lambda: sep(d1, w1, 1)) + (timed(lambda: sep(db, wb, 3)) if two else 0.0)
        row = ["%dx%dx%d %s: separate %.1f us" % (H, H, c, "3+3+1" if two else "3+1", t_sep)]
        for cfg in ("4,256,1024,1500000", "5,256,1024,1500000", "8,256,1024,1500000", "4,512,1024,1500000", "8,512,1024,1500000", "3,256,1024,1500000", "6,256,1024,1500000",
                    "4,256,2048,3000000", "8,256,2048,3000000"):
            os.environ["MAF_DWWG31"] = cfg
            f = lambda: lib.check(L.maf_dw_wgrad31(x.data_ptr(), c, da.data_ptr(), c, db.data_ptr() if two else None, c if two else 0, d1.data_ptr(), c, B, H, H, c, lib.F16,
                                                   wa.data_ptr(), wb.data_ptr() if two else None, w1.data_ptr(), 1, st))
            try:
                row.append("[%s] %.1f" % (cfg.split(",1")[0] + "/" + cfg.split(",")[2], timed(f)))
            except lib.MafError as e:
                row.append("[%s] n/a" % cfg)
        print("; ".join(row), flush=True)


if __name__ == "__main__":
    main()
