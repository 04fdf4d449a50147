#!/usr/bin/env python3
"""Measure the gap between the fp16 engine and the fp32 reference fixtures (what tests/test_gpu_model.py:_close16 bounds).
Prints max |d box| (px), max |d box| / max(|ref box|, 1) and max |d score| per case; the test tolerance is set to ~2x these."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maf_yolo_amd as M                      # noqa: E402
from oracle import maf_oracle as O            # noqa: E402

dev = torch.device("cuda:0")


def gap(p, ref, tag):
    d = np.abs(p[..., :4] - ref[..., :4])
    rel = d / np.maximum(np.abs(ref[..., :4]), 1.0)
    # the smallest atol that passes together with rtol = 5e-3
    need = (d - 5e-3 * np.abs(ref[..., :4])).max()
    print("%-28s box max %.3f px  rel max %.2e  atol needed at rtol 5e-3: %.3f  score max %.2e" % (tag, d.max(), rel.max(), need, np.abs(p[..., 4:] - ref[..., 4:]).max()))


for s in "nsm":
    g = np.load(os.path.join(ROOT, "tests", "golden", "maf_%s.npz" % s))
    m = M.Model(s)
    m.load_state_dict(O.synth_state_dict(s, 0))
    m = m.to(dev).eval()
    with torch.no_grad():
        p = m(O.synth_images(1, 320, 1).to(dev).half())[0].cpu().numpy()
    gap(p, g["pred320_deploy"], "%s 320 golden" % s)
    if s == "n":
        with torch.no_grad():
            p = m(O.synth_images(2, 640, 1).to(dev).half())[0][:, ::16].cpu().numpy()
        gap(p, g["pred640_rows16"], "n 640 golden rows16")
        for hw in ((384, 640), (352, 608), (64, 96)):
            x = torch.rand(2, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(hw[0] + hw[1]))
            ref = O.predict(O.reparam(O.synth_state_dict("n", 0), "n"), "n", x).numpy()
            with torch.no_grad():
                p = m(x.to(dev).half())[0].cpu().numpy()
            gap(p, ref, "n rect %dx%d" % hw)
        for fuse in (True, 2, False):
            m2 = M.Model("n"); m2.load_state_dict(O.synth_state_dict("n", 0)); m2 = m2.to(dev).eval(); m2.fuse_bottlenecks = fuse
            with torch.no_grad():
                p = m2(O.synth_images(2, 320, 8).to(dev).half())[0].cpu().numpy()
            ref = O.predict(O.reparam(O.synth_state_dict("n", 0), "n"), "n", O.synth_images(2, 320, 8)).numpy()
            gap(p, ref, "n 320 fuse=%s" % fuse)
