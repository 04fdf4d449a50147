#!/usr/bin/env python3
"""BatchNorm(train)+activation kernels of csrc/bn_act.hip in isolation: forward (statistics + apply) and backward (statistics + apply) of the shapes a
training step of MAF-YOLO-n runs at batch 32, with HIP events around each call.  Prints us per call and GB/s (3 / 5 passes of the tensor).

    python tools/bn_bench.py [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maf_yolo_amd import train_ops          # noqa: E402

SHAPES = [(32, 24, 320, 320), (32, 48, 160, 160), (32, 72, 160, 160), (32, 144, 80, 80), (32, 192, 80, 80), (32, 128, 80, 80), (32, 288, 40, 40), (32, 192, 40, 40),
          (32, 128, 40, 40), (32, 576, 20, 20), (32, 384, 20, 20), (32, 192, 20, 20), (32, 96, 20, 20)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    tot_f = tot_b = 0.0
    for B, C, H, W in SHAPES:
        x = torch.randn(B, C, H, W, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).to(dev).train()
        dy = torch.randn_like(x)
        for _ in range(3):
            y = train_ops.bn_act(x, bn, "silu")
            y.backward(dy)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(reps):
            x.grad = None
            e[0].record()
            y = train_ops.bn_act(x, bn, "silu")
            e[1].record()
            y.backward(dy)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        tf, tb = 1e3 * tf / reps, 1e3 * tb / reps
        nbytes = x.numel() * 2
        print("%-22s %7.1f MB  fwd %7.1f us %6.0f GB/s   bwd %7.1f us %6.0f GB/s" % ((B, C, H, W), nbytes / 1e6, tf, 3 * nbytes / tf / 1e3, tb, 5 * nbytes / tb / 1e3))
        tot_f += tf; tot_b += tb
    print("sum over the shapes: fwd %.1f us, bwd %.1f us" % (tot_f, tot_b))


if __name__ == "__main__":
    main()
