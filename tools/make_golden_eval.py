#!/usr/bin/env python3
"""Golden vectors for the evaluation loop ("what the build's own counterpart of the Python callers must reproduce"): the
reference's OWN deploy-form model, non_max_suppression and Evaler.convert_to_coco_format driven exactly as Evaler.predict_model drives
them (yolov6/core/evaler.py:157-187) — uint8 batch -> float / 255 -> model(imgs)[0] -> NMS(conf 0.03, iou 0.65, multi_label) -> COCO rows —
run here in the build container (CPU, fp32; the greedy-NMS inner kernel is oracle.greedy_nms_torch, see tools/ref_import.py).

    python tools/make_golden_eval.py    ->  tests/golden/eval_loop.npz

Inputs are regenerated from seeds by the test (`batches()` below is imported by it); only results are stored.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SIZE = 160
COCO_IDS = list(range(1, 12)) + list(range(13, 26)) + [27, 28] + list(range(31, 45)) + list(range(46, 66)) + [67, 70] + list(range(72, 83)) + [84, 85, 86, 87, 88, 89, 90]


def batches():
    """Two batches (3 + 2 images) as the reference's data loader yields them: (uint8 imgs [B,3,H,W], targets, paths, shapes)."""
    rs = np.random.RandomState(11)
    out = []
    img_no = 139
    for B in (3, 2):
        imgs = torch.from_numpy(rs.randint(0, 256, (B, 3, SIZE, SIZE)).astype(np.uint8))
        shapes, paths = [], []
        for b in range(B):
            h0, w0 = [(480, 640), (427, 640), (1080, 1920), (333, 500), (640, 480)][(img_no + b) % 5]
            r = min(SIZE / h0, SIZE / w0)
            nh, nw = int(round(h0 * r)), int(round(w0 * r))
            shapes.append(((h0, w0), ((nh / h0, nw / w0), ((SIZE - nw) / 2, (SIZE - nh) / 2))))
            paths.append("/data/coco/images/val2017/%012d.jpg" % (img_no + b))
        img_no += B
        out.append((imgs, torch.zeros(0, 6), paths, shapes))
    return out


def main():
    import ref_import
    import make_golden_post
    from oracle import maf_oracle as O
    torch.set_num_threads(os.cpu_count())
    Evaler = make_golden_post.load_evaler()                 # installs a dummy NMS stub ...
    ns = ref_import.load(O.greedy_nms_torch)                # ... replaced here by the greedy-NMS restatement
    model = ref_import.build(ns, "n")
    model.load_state_dict(O.synth_state_dict("n", seed=0, cls_bias=-3.0), strict=True)      # enough candidates above conf 0.03 on random images
    deploy = ref_import.to_deploy(ns, model.eval())
    ev = types.SimpleNamespace(scale_exact=False, is_coco=True)
    ev.scale_coords = types.MethodType(Evaler.scale_coords, ev)
    ev.box_convert = types.MethodType(Evaler.box_convert, ev)
    res, counts = [], []
    with torch.no_grad():
        for imgs, targets, paths, shapes in batches():
            x = imgs.float()
            x /= 255                                                             # evaler.py:161-163
            outputs, _ = deploy(x)                                               # :168
            outputs = ns.non_max_suppression(outputs, 0.03, 0.65, multi_label=True)     # :178
            counts += [int(o.shape[0]) for o in outputs]
            res.extend(Evaler.convert_to_coco_format(ev, outputs, x, paths, shapes, COCO_IDS))   # :187
    blob = {"counts": np.asarray(counts, np.int32),
            "image_id": np.asarray([r["image_id"] for r in res], np.int64), "category_id": np.asarray([r["category_id"] for r in res], np.int64),
            "bbox": np.asarray([r["bbox"] for r in res], np.float64).reshape(-1, 4), "score": np.asarray([r["score"] for r in res], np.float64)}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eval_loop.npz"), **blob)
    print("wrote eval_loop.npz: detections per image", counts, "rows", len(res))


if __name__ == "__main__":
    main()
