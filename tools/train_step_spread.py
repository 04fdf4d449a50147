#!/usr/bin/env python3
"""Run-to-run spread of ONE train step (fp32 or AMP) of the train-form graph on the GPU: the step is repeated N times on the same weights, images and
labels with the label assignment frozen; every parameter's gradient is compared with the element-wise MEDIAN over the runs.  Prints the worst
deviation per run and, for runs that stand out, the parameters that moved and whether the forward outputs moved too — the tool that located the
"one run in 40" outlier of tests/test_gpu_train.py.

    python tools/train_step_spread.py m atss 60 [amp|fp32] [det]      (det: bit-reproducible BatchNorm statistics, train_ops.set_deterministic)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maf_yolo_amd as M                    # noqa: E402
from maf_yolo_amd import synth              # noqa: E402

TARGETS = [[0, 3, 0.40, 0.50, 0.30, 0.40], [0, 17, 0.70, 0.30, 0.20, 0.50], [0, 17, 0.25, 0.75, 0.30, 0.25],
           [1, 5, 0.50, 0.50, 0.60, 0.60], [1, 62, 0.20, 0.30, 0.25, 0.35]]


def main():
    scale, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    amp = len(sys.argv) > 4 and sys.argv[4] == "amp"
    if "det" in sys.argv[4:]:
        from maf_yolo_amd import train_ops
        train_ops.set_deterministic(True)
    dev = torch.device("cuda:0")
    epoch, kw = (5, dict(warmup_epoch=0)) if tag == "tal" else (0, dict())
    sd = synth.synth_state_dict(M.Model(scale), scale, 0)
    x = synth.synth_images(2, 128, 7).to(dev)
    targets = torch.tensor(TARGETS, dtype=torch.float32, device=dev)
    runs, heads, frozen = [], [], None
    names = None
    for it in range(n):
        m = M.Model(scale)
        m.load_state_dict(sd)
        m = m.to(dev).train()
        crit = M.ComputeLoss(ori_img_size=128, **kw)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            (feats, cls, reg), _ = m(x)
        loss, items = crit((feats, cls, reg), targets, epoch, 1, assignment=frozen)
        if frozen is None:
            frozen = tuple(t.clone() for t in crit.last_assignment)
        (loss * (1024.0 if amp else 1.0)).backward()
        torch.cuda.synchronize()
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        runs.append([p.grad.detach().float().cpu().numpy().ravel() / (1024.0 if amp else 1.0) for k, p in m.named_parameters() if p.grad is not None])
        heads.append((cls.detach().float().cpu().numpy(), reg.detach().float().cpu().numpy(), float(loss)))
    med = [np.median(np.stack([r[i] for r in runs]), 0) for i in range(len(names))]
    gmax = max(float(np.abs(v).max()) for v in med)
    scale_ = [max(float(np.abs(v).max()), 1e-3 * gmax) for v in med]          # (a bias in front of a BatchNorm has a zero gradient in exact arithmetic: floor)
    dev_ = np.array([[float(np.abs(r[i] - med[i]).max()) / scale_[i] for i in range(len(names))] for r in runs])       # [run][param]
    worst = dev_.max(1)
    typical = float(np.median(worst))
    print("%s %s amp=%s: %d runs; worst |g - median| / max|g| per run: median %.2e, max %.2e" % (scale, tag, amp, n, typical, worst.max()))
    for it in np.argsort(-worst)[:6]:
        top = np.argsort(-dev_[it])[:6]
        dc = float(np.abs(heads[it][0] - heads[0][0]).max()); dr = float(np.abs(heads[it][1] - heads[0][1]).max())
        print("  run %3d: worst %.2e; head outputs vs run 0: d cls %.1e d reg %.1e d loss %.1e; " % (it, worst[it], dc, dr, abs(heads[it][2] - heads[0][2]))
              + ", ".join("%s %.1e" % (names[i], dev_[it][i]) for i in top))
    out = worst > max(10 * typical, 1e-2)
    print("outlier runs:", np.nonzero(out)[0].tolist())


if __name__ == "__main__":
    main()
