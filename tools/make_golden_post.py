"""Golden vectors for the post-NMS tail: the reference's own Evaler.scale_coords / box_convert /
convert_to_coco_format (yolov6/core/evaler.py:374-434) run here, in the build container, on seeded detections.

    python tools/make_golden_post.py        ->  tests/golden/post_cases.npz

Only the three pure-PyTorch methods are exercised; pycocotools and the data loader that evaler.py imports at module level are
stubbed (they are not on this path)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402


def load_evaler():
    ref_import.load(lambda b, s, t: torch.zeros(0, dtype=torch.long))          # installs the cv2 / torchvision / timm / addict stubs
    for name, attrs in (("pycocotools", {}), ("pycocotools.coco", {"COCO": object}), ("pycocotools.cocoeval", {"COCOeval": object}),
                        ("yolov6.data.data_load", {"create_dataloader": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.path.insert(0, ref_import.REF)
    from yolov6.core.evaler import Evaler
    return Evaler


def cases():
    g = torch.Generator().manual_seed(1234)
    out = []
    for ci, (scale_exact, B) in enumerate([(False, 3), (True, 3), (False, 1)]):
        outputs, shapes, paths = [], [], []
        for b in range(B):
            n = [0, 7, 300, 41][(b + ci) % 4] if not (ci == 2) else 5
            h0, w0 = [(480, 640), (427, 640), (1080, 1920), (333, 500)][(b + ci) % 4]
            r = min(640 / h0, 640 / w0)
            nh, nw = int(round(h0 * r)), int(round(w0 * r))
            pad = ((640 - nw) / 2, (640 - nh) / 2)
            xy = torch.rand(n, 2, generator=g) * 700 - 30                        # some boxes stick out of the image: clamp path
            wh = torch.rand(n, 2, generator=g) * 300 + 1
            det = torch.cat([xy, xy + wh, torch.rand(n, 1, generator=g), torch.randint(0, 80, (n, 1), generator=g).float()], 1)
            outputs.append(det)
            shapes.append(((h0, w0), ((nh / h0, nw / w0), pad)))
            paths.append("/data/coco/images/val2017/%012d.jpg" % (139 + 1000 * b + ci))
        out.append((scale_exact, outputs, shapes, paths))
    return out


def main():
    Evaler = load_evaler()
    ids = list(range(1, 12)) + list(range(13, 26)) + [27, 28] + list(range(31, 45)) + list(range(46, 66)) + [67, 70] + \
        list(range(72, 83)) + [84, 85, 86, 87, 88, 89, 90]                       # the COCO 80 -> 91 table (evaler.py coco80_to_coco91_class)
    assert len(ids) == 80
    blob = {"ids": np.asarray(ids, np.int32)}
    for ci, (scale_exact, outputs, shapes, paths) in enumerate(cases()):
        ev = types.SimpleNamespace(scale_exact=scale_exact, is_coco=True)
        ev.scale_coords = types.MethodType(Evaler.scale_coords, ev)
        ev.box_convert = types.MethodType(Evaler.box_convert, ev)
        imgs = torch.zeros(len(outputs), 3, 640, 640)
        res = Evaler.convert_to_coco_format(ev, [o.clone() for o in outputs], imgs, paths, shapes, ids)
        blob["c%d_scale_exact" % ci] = np.asarray(int(scale_exact))
        blob["c%d_counts" % ci] = np.asarray([o.shape[0] for o in outputs], np.int32)
        blob["c%d_dets" % ci] = torch.cat(outputs, 0).numpy()
        blob["c%d_shapes" % ci] = np.asarray([[s[0][0], s[0][1], s[1][0][0], s[1][0][1], s[1][1][0], s[1][1][1]] for s in shapes], np.float64)
        blob["c%d_image_ids" % ci] = np.asarray([int(os.path.splitext(os.path.basename(p))[0]) for p in paths], np.int64)
        blob["c%d_out_image_id" % ci] = np.asarray([r["image_id"] for r in res], np.int64)
        blob["c%d_out_category_id" % ci] = np.asarray([r["category_id"] for r in res], np.int64)
        blob["c%d_out_bbox" % ci] = np.asarray([r["bbox"] for r in res], np.float64).reshape(-1, 4)
        blob["c%d_out_score" % ci] = np.asarray([r["score"] for r in res], np.float64)
        print("case", ci, "rows", len(res))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "post_cases.npz"), **blob)


if __name__ == "__main__":
    main()
