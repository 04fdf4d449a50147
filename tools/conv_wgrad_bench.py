"""Every dense weight-gradient launch of a MAF-YOLO-n training step at batch 32 (shapes from profiles/round5_train_launches_isolated.md), each alone:
microseconds (HIP events, 20 launches), GB/s of its algorithmic bytes (x + dY once), and the deviation from the framework's fp32 weight gradient of the same fp16
tensors in units of max |g|.      python tools/conv_wgrad_bench.py [check]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maf_yolo_amd import lib          # noqa: E402

# (input map, Cin, Cout, k, stride, launches per step)
SHAPES = [(640, 8, 24, 3, 2, 1), (320, 24, 48, 3, 2, 1), (640, 8, 24, 1, 2, 1), (320, 24, 48, 1, 2, 1), (160, 48, 64, 3, 2, 1), (160, 48, 48, 3, 2, 1), (160, 48, 48, 1, 2, 1),
          (80, 128, 128, 3, 2, 2), (80, 96, 96, 3, 2, 1), (80, 96, 64, 3, 2, 1), (80, 96, 96, 1, 2, 1), (40, 192, 192, 3, 2, 1), (40, 192, 96, 3, 2, 1), (40, 128, 128, 3, 2, 2), (40, 192, 192, 1, 2, 1),
          (160, 72, 48, 1, 1, 1), (160, 24, 72, 1, 1, 1), (160, 72, 24, 1, 1, 1), (160, 48, 48, 1, 1, 1),
          (80, 288, 128, 1, 1, 1), (80, 256, 128, 1, 1, 1), (80, 144, 96, 1, 1, 1), (80, 192, 128, 1, 1, 2), (80, 96, 96, 1, 1, 1), (80, 128, 80, 1, 1, 1), (80, 48, 144, 1, 1, 1), (80, 128, 72, 1, 1, 1),
          (80, 64, 192, 1, 1, 2), (80, 128, 128, 1, 1, 3), (80, 192, 64, 1, 1, 2), (80, 144, 48, 1, 1, 1), (80, 48, 48, 1, 1, 1),
          (40, 576, 128, 1, 1, 1), (40, 448, 128, 1, 1, 1), (40, 288, 192, 1, 1, 1), (40, 288, 96, 1, 1, 1), (40, 192, 192, 1, 1, 1), (40, 192, 128, 1, 1, 2), (40, 96, 288, 1, 1, 1), (40, 192, 64, 1, 1, 2),
          (40, 128, 128, 1, 1, 3), (40, 128, 80, 1, 1, 1), (40, 128, 72, 1, 1, 1), (40, 96, 96, 1, 1, 1), (40, 64, 192, 1, 1, 2),
          (20, 768, 384, 1, 1, 1), (20, 576, 384, 1, 1, 1), (20, 576, 192, 1, 1, 1), (20, 480, 192, 1, 1, 1), (20, 384, 384, 1, 1, 1), (20, 448, 192, 1, 1, 1), (20, 384, 192, 1, 1, 1), (20, 288, 192, 1, 1, 2),
          (20, 192, 576, 1, 1, 1), (20, 288, 96, 1, 1, 2), (20, 192, 192, 1, 1, 4), (20, 96, 288, 1, 1, 2), (20, 192, 72, 1, 1, 1), (20, 192, 80, 1, 1, 1)]


def main():
    check = len(sys.argv) > 1
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B = 32
    tot = {}
    print("| x map | Cin | Cout | k | s | us | GB/s | err |\n|---|---|---|---|---|---|---|---|")
    for (H, cin, cout, k, s, n) in SHAPES:
        Ho = (H - 1) // s + 1
        g = torch.Generator().manual_seed(H + cin + cout)
        x = torch.randn(B, H, H, cin, generator=g).half().cuda()
        dy = torch.randn(B, Ho, Ho, cout, generator=g).half().cuda()
        dw = torch.zeros(k, k, cout, cin, device="cuda")
        f = lambda: lib.check(L.maf_conv_wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, B, Ho, Ho, H, H, cin, cout, k, s, lib.F16, dw.data_ptr(), st))
        err = float("nan")
        if check and H <= 160:
            f()
            torch.cuda.synchronize()
            ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (cout, cin, k, k), dy.permute(0, 3, 1, 2).float(), stride=s, padding=k // 2)
            err = float((dw.permute(2, 3, 0, 1) - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nb = (x.numel() + dy.numel()) * 2
        print("| %d | %d | %d | %d | %d | %.1f | %.0f | %.1e |" % (H, cin, cout, k, s, us, nb / us / 1e3, err), flush=True)
        key = "3x3 s2" if k == 3 else "1x1 s2" if s == 2 else "1x1 %d" % H
        tot[key] = tot.get(key, 0.0) + n * us
    print("\nper step (launch counts of MAF-YOLO-n applied): " + ", ".join("%s: %.0f us" % kv for kv in tot.items()) + "; all: %.0f us" % sum(tot.values()))


if __name__ == "__main__":
    main()
