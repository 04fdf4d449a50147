"""`non_max_suppression` — the reference's post-processing surface (yolov6/utils/nms.py:31-105) on the GPU.

Same signature, same return type (a Python list of [n_i, 6] tensors (x1,y1,x2,y2,conf,cls) on
prediction.device, `zeros((0,6))`-shaped for empty images), same AssertionError on bad thresholds
(nms.py:50-51).  The whole batch is two kernel launches (csrc/nms.hip) and ONE device->host copy
(the per-image counts) instead of a Python loop with ~6 implicit syncs per image.

Differences, all documented in DESIGN.md: decode/NMS arithmetic is fp32 even for fp16 predictions
(the reference's fp16 `cls*4096` overflows for cls >= 16); the 10 s time limit
(nms.py:101-103) is dropped; max_det <= 1024.
"""
import torch

from . import lib

_ws = {}


def nms_raw(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """Device-side result without the host sync: (rows [B,max_det,6], idx int64 [B,max_det], count int32 [B])."""
    assert 0 <= conf_thres <= 1, f'conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided.'
    assert 0 <= iou_thres <= 1, f'iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided.'
    if not prediction.is_cuda:
        raise lib.MafError("non_max_suppression runs on the HIP path only: got a %s tensor (no CPU fallback)" % prediction.device)
    pred = prediction
    if pred.dtype != torch.float32:
        pred = pred.float()
    pred = pred.contiguous()
    B, N, no = pred.shape
    nc = no - 5
    dev = pred.device
    L = lib.load()
    need = L.maf_nms_workspace_bytes(B, N, nc)
    key = (dev.index, B, N, nc)
    ws = _ws.get(key)
    if ws is None or ws.numel() < need:
        if len(_ws) > 4:
            _ws.clear()
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _ws[key] = ws
    rows = torch.zeros(B, max_det, 6, dtype=torch.float32, device=dev)
    idx = torch.zeros(B, max_det, dtype=torch.int64, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    cls_t, ncls = None, 0
    if classes is not None:
        cls_t = torch.as_tensor(list(classes), dtype=torch.int32, device=dev)
        ncls = cls_t.numel()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        lib.check(L.maf_nms(pred.data_ptr(), B, N, nc, float(conf_thres), float(iou_thres),
                            cls_t.data_ptr() if ncls else None, ncls, int(bool(agnostic)), int(bool(multi_label)),
                            int(max_det), ws.data_ptr(), ws.numel(), rows.data_ptr(), idx.data_ptr(), cnt.data_ptr(), stream))
    return rows, idx, cnt


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        max_det=300, return_index=False):
    rows, idx, cnt = nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
    counts = cnt.tolist()                              # the only device->host sync
    out = [rows[b, :n] for b, n in enumerate(counts)]
    if return_index:
        return out, [idx[b, :n] for b, n in enumerate(counts)]
    return out
