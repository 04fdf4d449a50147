"""Post-NMS tail of the evaluation loop on the GPU.

`convert_to_coco_format(outputs, imgs, paths, shapes, ids)` has the signature and the return value of
Evaler.convert_to_coco_format (yolov6/core/evaler.py:411-434): a list of
{"image_id", "category_id", "bbox": [x, y, w, h], "score"} dicts, boxes rescaled to the original image with
Evaler.scale_coords (:382-409) and rounded to 3 / 5 decimals.  The reference loops over boxes in Python with a
`.tolist()` / `.item()` per box; here the batch is one kernel (csrc/post.hip) and ONE device->host copy, the decimal
rounding is one numpy call.  `outputs` is either the reference's list of [n_i, 6] tensors or — to skip the host round trip of
the NMS result as well — the (rows, idx, count) triple of nms.nms_raw / an NmsHandle.
"""
from pathlib import Path

import numpy as np
import torch

from . import lib
from .nms import NmsHandle


def coco_rows(rows, count, shapes, ids, scale_exact=False, stream=None):
    """Device part.  rows [B,max_det,6] fp32, count [B] int32 (cuda) -> (packed [R,7] fp32 cuda tensor, R)."""
    if not rows.is_cuda:
        raise lib.MafError("coco_rows runs on the HIP path only: got a %s tensor (no CPU fallback)" % rows.device)
    B, max_det, _ = rows.shape
    dev = rows.device
    par = np.empty((B, 6), np.float32)
    for i, s in enumerate(shapes):
        (h0, w0), (gain, pad) = s[0], s[1]
        par[i] = (h0, w0, gain[1] if scale_exact else gain[0], gain[0], pad[0], pad[1])
    par_t = torch.from_numpy(par).to(dev)
    ids_t = torch.as_tensor(np.asarray(ids, np.int32), device=dev) if ids is not None and len(ids) else None
    out = torch.empty(B * max_det, 7, dtype=torch.float32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    lib.check(lib.load().maf_coco_rows(rows.contiguous().data_ptr(), count.data_ptr(), B, max_det, par_t.data_ptr(),
                                       ids_t.data_ptr() if ids_t is not None else None, 0 if ids_t is None else ids_t.numel(),
                                       out.data_ptr(), total.data_ptr(), st.cuda_stream))
    return out, total


def convert_to_coco_format(outputs, imgs, paths, shapes, ids, is_coco=True, scale_exact=False):
    if isinstance(outputs, NmsHandle):
        outputs.event.synchronize()
        rows, count = outputs.rows, outputs.cnt
    elif isinstance(outputs, tuple) and len(outputs) == 3 and torch.is_tensor(outputs[2]):
        rows, _, count = outputs
    else:                                                   # the reference's list of [n_i, 6] tensors
        B = len(outputs)
        md = max(1, max(int(o.shape[0]) for o in outputs))
        dev = outputs[0].device
        rows = torch.zeros(B, md, 6, dtype=torch.float32, device=dev)
        for b, o in enumerate(outputs):
            rows[b, :o.shape[0]] = o.float()
        count = torch.tensor([int(o.shape[0]) for o in outputs], dtype=torch.int32, device=dev)
    packed, total = coco_rows(rows, count, shapes, ids, scale_exact)
    n = int(total.item())                                   # the one host sync
    arr = packed[:n].cpu().numpy().astype(np.float64)
    bbox = np.round(arr[:, 2:6] * 1000.0) / 1000.0          # == round(v, 3) for doubles that come from fp32: v * 1000 is exact
    score = np.round(arr[:, 6] * 100000.0) / 100000.0
    image_ids = [int(Path(p).stem) if is_coco else Path(p).stem for p in paths]
    res = []
    for r in range(n):
        res.append({"image_id": image_ids[int(arr[r, 0])], "category_id": int(arr[r, 1]),
                    "bbox": bbox[r].tolist(), "score": float(score[r])})
    return res
