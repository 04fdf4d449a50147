#!/usr/bin/env python3
"""Tile sweep of the merged depth-wise branch kernels (csrc/dw_branches.hip): for the block shapes of a training step of MAF-YOLO-n at batch 32, every
(rows, cols, channels) tile through MAF_DWB_TILE / MAF_DWB_TILE_DGRAD, forward and summed data gradient, HIP events.  Prints the best tiles and what the
built-in cost model picks.     python tools/dwb_sweep.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maf_yolo_amd import train_ops          # noqa: E402

SHAPES = [(160, 72, 3), (80, 144, 5), (80, 192, 5), (80, 128, 5), (40, 288, 7), (40, 192, 7), (40, 128, 7), (20, 576, 9), (20, 288, 9), (20, 192, 9)]
KS = {3: (3, 3, 1), 5: (5, 3, 1), 7: (7, 5, 3), 9: (9, 7, 5, 3)}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    B = 32
    for H, C, k0 in SHAPES:
        x = torch.randn(B, C, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        ws = [torch.randn(C, 1, k, k, device=dev) / k for k in KS[k0]]
        dzs = [torch.randn_like(x) for _ in ws]
        outs = [torch.empty_like(x) for _ in ws]
        dx = torch.empty_like(x)
        wf = [train_ops._packed_dw(w, C, w.shape[-1], 0, train_ops.lib.F16, dev) for w in ws]
        wb = [train_ops._packed_dw(w, C, w.shape[-1], 1, train_ops.lib.F16, dev) for w in ws]

        def run(dgrad):
            if dgrad:
                train_ops._launch_dwb(dzs, [dx], wb, k0, B, H, H, C, train_ops.lib.F16, True, dev)
            else:
                train_ops._launch_dwb([x], outs, wf, k0, B, H, H, C, train_ops.lib.F16, False, dev)

        def timed(dgrad):
            for _ in range(2):
                run(dgrad)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(dgrad)
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / reps
        for dgrad in (False, True):
            var = "MAF_DWB_TILE_DGRAD" if dgrad else "MAF_DWB_TILE"
            os.environ.pop(var, None)
            base = timed(dgrad)
            res = []
            for th in (4, 5, 8, 10, 16, 20, 32, 40):
                for tw in (8, 16, 20, 32, 40):
                    for cb in (16, 32, 64):
                        if th > H or tw > H or H % th or (H % tw and tw not in (16, 32)):
                            continue
                        os.environ[var] = "%d,%d,%d" % (th, tw, cb)
                        res.append((timed(dgrad), th, tw, cb))
            os.environ.pop(var, None)
            res.sort()
            print("%3d x %3d x %3d k%d %s: built-in %.1f us; best %s" % (H, H, C, k0, "dgrad" if dgrad else "fwd  ", base,
                  ", ".join("%.1f us (%d,%d,%d)" % r for r in res[:4])), flush=True)


if __name__ == "__main__":
    main()
