#!/usr/bin/env python3
"""BASELINE configs[0] at its stated batch: MAF-YOLO-n, 8 x 3 x 640 x 640, the reference's own deploy-form forward + non_max_suppression on the CPU
(tools/eval.py --device cpu: evaler.py:160-180 — model(imgs)[0], then NMS at conf 0.03 / IoU 0.65 / multi_label).

    python tools/make_golden_b8.py      ->  tests/golden/maf_n_b8.npz

Run in the build container only (imports /root/reference through tools/ref_import.py); the fixture holds DATA: every 64th anchor row of the [8, 8400, 85]
prediction, float64 column sums per image (a checksum over all 8 400 anchors), the detection rows of all eight images and their counts.  The input batch is
regenerated from its seed by the tests (oracle.maf_oracle.synth_images(8, 640, seed=1): images 0 and 1 are the two of maf_n.npz's B = 2 fixture)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import                      # noqa: E402
from oracle import maf_oracle as O     # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    ns = ref_import.load(O.greedy_nms_torch)
    model = ref_import.build(ns, "n")
    model.load_state_dict(O.synth_state_dict("n", seed=0), strict=True)
    model.eval()
    deploy = ref_import.to_deploy(ns, model)
    x = O.synth_images(8, 640, seed=1)
    with torch.no_grad():
        pred, _ = deploy(x)
    assert pred.shape == (8, 8400, 85)
    rec = {"pred640_b8_rows64": pred[:, ::64].numpy(), "pred640_b8_colsum": pred.double().sum(1).numpy()}
    dets = ns.non_max_suppression(pred.clone(), conf_thres=0.03, iou_thres=0.65, multi_label=True)
    rec["nms640_b8_n"] = np.array([d.shape[0] for d in dets])
    for bi, d in enumerate(dets):
        rec["nms640_b8_%d" % bi] = d.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "maf_n_b8.npz"), **rec)
    print("wrote maf_n_b8.npz", {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
