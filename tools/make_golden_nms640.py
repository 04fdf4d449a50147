#!/usr/bin/env python3
"""Reference detections at the BASELINE size for the s and m graphs (build container only): the REFERENCE's deploy-form Model (fuse_model /
switch_to_deploy / reparameterize) on 2 x 3x640x640 seeded images, then its own non_max_suppression(0.03, 0.65, multi_label) —
what tools/make_golden.py stores as `nms640_eval_*` for n — as DATA for tests/test_gpu_fused_parity.py (fp16 engine -> NMS vs the
reference's fp32 detections, VERDICT r2 task 4d).

    python tools/make_golden_nms640.py s m    ->  tests/golden/nms640_s.npz, nms640_m.npz

The inner torchvision.ops.nms is the oracle's restatement (torchvision is absent: SURVEY.md §8c)."""
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
    ns = ref_import.load(O.greedy_nms_torch)
    for scale in [a for a in sys.argv[1:] if a in ("n", "s", "m")] or ["s", "m"]:
        model = ref_import.build(ns, scale)
        model.load_state_dict(O.synth_state_dict(scale, seed=0), strict=True)
        model.eval()
        deploy = ref_import.to_deploy(ns, model)
        x = O.synth_images(2, 640, seed=1)
        with torch.no_grad():
            p640, _ = deploy(x)
        rec = {"pred640_rows16": p640[:, ::16].numpy(), "pred640_colsum": p640.double().sum(1).numpy()}
        dets = ns.non_max_suppression(p640.clone(), conf_thres=0.03, iou_thres=0.65, multi_label=True)
        for bi, d in enumerate(dets):
            rec["nms640_eval_%d" % bi] = d.numpy()
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nms640_%s.npz" % scale), **rec)
        print("wrote nms640_%s.npz:" % scale, [tuple(d.shape) for d in dets], "candidates", int((p640[..., 5:] > 0.03).sum()))


if __name__ == "__main__":
    main()
