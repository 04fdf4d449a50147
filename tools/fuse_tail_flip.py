#!/usr/bin/env python3
"""Which detections change when the closing 1x1 conv of a RepHDW block runs inside the block's fused bottleneck launch (Model.fuse_tail) — VERDICT r4 #4(a).

For scale s / m at 2 x 640 x 640 (the images of tests/golden/nms640_<scale>.npz): the fp16 engine with fuse_tail off and on, NMS(0.03, 0.65, multi_label) with the flat
survivor indices, both matched against the reference's fp32 detections (tests/test_gpu_fused_parity.match_detections).  For every reference row that one plan
matches and the other does not, the raw candidate (anchor, class) is looked up in BOTH plans' predictions and its IoU with the box that suppressed it (the kept
same-class detection of highest IoU) is printed for both plans: an IoU that sits on either side of the 0.65 threshold by ~1e-3 is the fp16-class event the n test
already tolerates (one per image), anything else is a defect of the fused kernel.

    python tools/fuse_tail_flip.py s|m
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import maf_yolo_amd as M                                    # noqa: E402
from maf_yolo_amd import lib                                # noqa: E402
from oracle import maf_oracle as O                          # noqa: E402  (test infrastructure: seeded weights / images)
from test_gpu_fused_parity import match_detections, _iou   # noqa: E402


def run(scale, fuse):
    dev = torch.device("cuda:0")
    m = M.Model(scale)
    m.load_state_dict(O.synth_state_dict(scale, 0))
    m = m.to(dev).eval()
    m.fuse_tail = fuse
    x = O.synth_images(2, 640, 1).to(dev).half()
    with torch.no_grad():
        pred = m(x)[0]
    plan = m.plan_for(x)
    ntail = sum(1 for o in plan.ops if o.kind == lib.OP_BOTTLENECK and o.nc > 0)
    dets, idx = M.non_max_suppression(pred, 0.03, 0.65, multi_label=True, return_index=True)
    return pred.float().cpu().numpy(), [d.cpu().numpy() for d in dets], [i.cpu().numpy() for i in idx], ntail, len(plan.ops)


def xyxy(p):
    return np.stack([p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2, p[:, 0] + p[:, 2] / 2, p[:, 1] + p[:, 3] / 2], 1)


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "s"
    g = np.load(os.path.join(ROOT, "tests", "golden", "nms640_%s.npz" % scale))
    res = {f: run(scale, f) for f in (False, True)}
    print("%s: launches %d (fuse_tail off) / %d (on: %d bottleneck launches carry the closing conv)" % (scale, res[False][4], res[True][4], res[True][3]))
    nc = res[False][0].shape[2] - 5
    for b in range(2):
        ref = g["nms640_eval_%d" % b]
        m_ = {}
        for f in (False, True):
            pairs, miss, extra = match_detections(res[f][1][b], ref)
            m_[f] = (dict(pairs), set(miss))
            print("image %d fuse_tail=%s: %d/%d matched, %d missed" % (b, f, len(pairs), ref.shape[0], len(miss)))
        for f_has, f_not in ((False, True), (True, False)):
            for i in sorted(m_[f_not][1] - m_[f_has][1]):
                j = m_[f_has][0][i]                                   # the row of plan f_has that matched reference row i
                flat = int(res[f_has][2][b][j])                       # its flat (anchor * nc + class) index
                a, c = flat // nc, flat % nc
                line = "  ref row %3d (class %2d, score %.4f): matched with fuse_tail=%s, not with %s; anchor %d" % (i, c, ref[i, 4], f_has, f_not, a)
                for f in (False, True):
                    p = res[f][0][b]
                    box = xyxy(p[a:a + 1, :4])[0]
                    sc = p[a, 4] * p[a, 5 + c] if p[a, 4] != 1 else p[a, 5 + c]
                    det = res[f][1][b]
                    same = det[(det[:, 5] == c) & (det[:, 4] >= sc - 1e-6)]
                    if same.size:
                        ious = _iou(box[None], same[:, :4])[0]
                        k = int(np.argmax(ious))
                        line += " | fuse_tail=%s: score %.5f, max IoU with a kept same-class box of higher score %.5f" % (f, sc, ious[k])
                    else:
                        line += " | fuse_tail=%s: score %.5f, no kept same-class box above it" % (f, sc)
                print(line)


if __name__ == "__main__":
    main()
