#!/usr/bin/env python3
"""Where the HOST time of one training step goes: cProfile over a few steady-state steps of bench.py's train leg (n, bs 32, AMP, GradExchange, fused SGD,
EMA), sorted by own time.  The step is issue-bound on the host, so this table is the to-do list.

    python tools/train_host_profile.py [steps]
"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maf_yolo_amd as M                    # noqa: E402
from maf_yolo_amd import synth              # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    model = M.Model("n")
    model.load_state_dict(synth.synth_state_dict(model, "n", 0))
    model = model.to(dev).train()
    ex = M.GradExchange(model)
    opt = M.build_optimizer(model, lr0=0.005, momentum=0.937, weight_decay=5e-4)
    scaler = torch.amp.GradScaler("cuda")
    ema = M.ModelEMA(model)
    B = 32
    x = synth.synth_images(B, 640, seed=1).to(dev)
    g = torch.Generator().manual_seed(100)
    wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
    targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
    crit = M.ComputeLoss(ori_img_size=640, warmup_epoch=0)

    def step():
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = model(x)
        loss = crit((feats, cls, reg), targets, 0, 0)[0]
        ex.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        ema.update(model)

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    # host issue time against an EMPTY queue: synchronise before every step, time until the last launch of the step has been issued
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("host issue time per step (empty queue): %s ms" % ", ".join("%.2f" % (1e3 * t) for t in ts))
    # the same, phase by phase (host clock only, no device synchronisation inside the step)
    acc = [0.0] * 6
    for _ in range(steps):
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = model(x)
        t.append(time.perf_counter())
        loss = crit((feats, cls, reg), targets, 0, 0)[0]
        ex.zero_grad()
        t.append(time.perf_counter())
        scaler.scale(loss).backward()
        t.append(time.perf_counter())
        scaler.step(opt)
        scaler.update()
        t.append(time.perf_counter())
        ema.update(model)
        t.append(time.perf_counter())
        for i in range(5):
            acc[i] += t[i + 1] - t[i]
    print("host issue time by phase (ms): forward %.2f, loss + zero_grad %.2f, backward %.2f, scaler + optimizer %.2f, EMA %.2f" % tuple(1e3 * a_ / steps for a_ in acc[:5]))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    print("step, back to back: %.2f ms" % (1e3 * (time.perf_counter() - t0) / steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(45)
    print("(cProfile inflates everything by its own overhead; per step = / %d)" % steps)
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
