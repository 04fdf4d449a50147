"""`non_max_suppression` — the reference's post-processing surface (yolov6/utils/nms.py:31-105) on the GPU.

Same signature, same return type (a Python list of [n_i, 6] tensors (x1,y1,x2,y2,conf,cls) on
prediction.device, `zeros((0,6))`-shaped for empty images), same AssertionError on bad thresholds
(nms.py:50-51).  The whole batch is five kernel launches (csrc/nms.hip) and ONE device->host copy
(the per-image counts) instead of a Python loop with ~6 implicit syncs per image.

Differences, all documented in DESIGN.md: decode/NMS arithmetic is fp32 even for fp16 predictions
(the reference's fp16 `cls*4096` overflows for cls >= 16); the 10 s time limit
(nms.py:101-103) is dropped; max_det <= 1024.
"""
import os

import torch

from .config import cfg
from . import lib

_ws = {}
_side = {}
# Which torchvision.ops.nms kernel the IoU comparison follows (yolov6/utils/nms.py:96 calls whichever matches the tensor's device):
#   "cpu"  (default) fp32 IoU > double threshold  — the rule of the oracle, of the golden fixtures and of an evaluation on the CPU
#   "cuda"           fp32 IoU > fl32(threshold)   — torchvision's GPU kernel takes `float iou_threshold`; differs from "cpu" only for a pair whose
#                    fp32 IoU equals fl32(thr) exactly with fl32(thr) > thr (0.6, 0.7 ...: the CPU rule suppresses it, the CUDA rule keeps it)
# Per call: non_max_suppression(..., iou_rule="cuda"); process-wide: maf_yolo_amd.nms.IOU_RULE = "cuda".
IOU_RULE = "cpu"


_cand_ws = {}


def candidate_workspace(dev, B, N, nc, slot=0):
    """NMS workspace that a forward pass fills with candidates (Model.nms_filter -> maf_engine_run_filtered) and the NMS call of that prediction
    then reads (MAF_NMS_PRECOLLECTED).  One per (device, shape, slot): the forward waits — on ITS stream — for the event the last NMS call that
    read this workspace recorded, so a slot's next forward cannot overwrite lists a still running NMS reads (the serving loop runs the NMS of
    batch i on a side stream while the forward of batch i + S reuses the slot)."""
    key = (dev.index, B, N, nc, slot)
    ws = _cand_ws.get(key)
    if ws is None:
        ws = _cand_ws[key] = torch.empty(lib.load().maf_nms_workspace_bytes(B, N, nc), dtype=torch.uint8, device=dev)
        ws._maf_busy = None
        ws._maf_gen = 0
    if ws._maf_busy is not None:
        torch.cuda.current_stream(dev).wait_event(ws._maf_busy)
    # every filtered forward that refills the workspace takes a new generation: a prediction carries the generation of ITS forward, and
    # the NMS uses the lists only while the workspace still holds that one (`p1 = model(x1); p2 = model(x2); nms(p1)` must not read
    # forward 2's lists for p1 — it falls back to its own pass over p1)
    ws._maf_gen += 1
    return ws


def _cand_valid(prediction, cand):
    """The candidate lists belong to this very tensor as its forward wrote it: same workspace generation, and nobody has written the tensor
    in place since (torch's version counter)."""
    return len(cand) == 4 and cand[0]._maf_gen == cand[2] and prediction._version == cand[3]


def _workspace(dev, st, B, N, nc, need):
    """Scratch buffer for one NMS call on stream `st`.  One buffer per (stream, shape): calls on the same stream are ordered, so they may
    share it; calls on different streams never do.  The buffer comes from the caching allocator while `st` is current, so when an entry is
    dropped the allocator hands the block out again only in `st`'s own order — nothing in flight can lose its workspace."""
    key = (dev.index, st.cuda_stream, B, N, nc)
    ws = _ws.get(key)
    if ws is None or ws.numel() < need:
        if len(_ws) >= 16:
            _ws.pop(next(iter(_ws)))                  # oldest entry; stream-ordered reuse keeps that safe (see above)
        ws = _ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws


def _matrix_form(B, overlapped):
    """Which form the all-pairs path of this call takes (same survivors bit for bit; csrc/nms.hip).  MATRIX_PATH True / False force one; "auto":
    * a call on a side stream under other work (non_max_suppression_async: the serving loop, the NMS of batch i under the forward of batch i + 1) takes the kept-list
      scan — one workgroup per image, B of the 256 CUs, ~128 us whatever B is; measured +3 % on the 3-in-flight loop against the matrix form, whose mask kernel takes
      the whole chip away from the forward for ~1.8 us per image;
    * a call with the GPU to itself (non_max_suppression: forward, then NMS, then the host — the reference's Evaler, the bs = 1 latency path) takes the matrix form up to
      32 images: the mask on the whole chip (1.8 us per image) + the row scan (~68 us) finish before the one-CU-per-image scan does."""
    if MATRIX_PATH == "auto":
        return (not overlapped) and B <= 32
    return bool(MATRIX_PATH)


def nms_raw(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300, stream=None, iou_rule=None):
    """Device-side result without the host sync: (rows [B,max_det,6], idx int64 [B,max_det], count int32 [B]).
    Launches on `stream` (a torch.cuda.Stream) or on the current stream."""
    assert 0 <= conf_thres <= 1, f'conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided.'
    assert 0 <= iou_thres <= 1, f'iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided.'
    if not prediction.is_cuda:
        raise lib.MafError("non_max_suppression runs on the HIP path only: got a %s tensor (no CPU fallback)" % prediction.device)
    B, N, no = prediction.shape
    nc = no - 5
    dev = prediction.device
    L = lib.load()
    with torch.cuda.device(dev):
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            # the fp16 -> fp32 cast (the reference's --half predictions) and the compaction of a strided view run on `st`, the stream
            # the kernels read them on (the caller has ordered `st` after the producer of `prediction`)
            pred = prediction
            if pred.dtype != torch.float32:
                pred = pred.float()
            pred = pred.contiguous()
            # candidate lists written by the forward pass that produced this very tensor (Model.nms_filter): usable if this call filters the same way
            cand = getattr(prediction, "_maf_cand", None)
            pre = (cand is not None and _cand_valid(prediction, cand) and pred is prediction and multi_label and nc > 1 and classes is None and float(conf_thres) == cand[1]
                   and not _single_launch(B) and cand[0].numel() >= L.maf_nms_workspace_bytes(B, N, nc))
            ws = cand[0] if pre else _workspace(dev, st, B, N, nc, L.maf_nms_workspace_bytes(B, N, nc))
            rows = torch.empty(B, max_det, 6, dtype=torch.float32, device=dev)
            idx = torch.empty(B, max_det, dtype=torch.int64, device=dev)
            cnt = torch.empty(B, dtype=torch.int32, device=dev)
            cls_t, ncls = None, 0
            if classes is not None:
                cls_t = torch.as_tensor(list(classes), dtype=torch.int32, device=dev)
                ncls = cls_t.numel()
            rule = iou_rule or IOU_RULE
            assert rule in ("cpu", "cuda"), "iou_rule: 'cpu' (double threshold) or 'cuda' (float threshold)"
            lib.check(L.maf_nms_ex(pred.data_ptr(), B, N, nc, float(conf_thres), float(iou_thres),
                                   cls_t.data_ptr() if ncls else None, ncls, int(bool(agnostic)), int(bool(multi_label)),
                                   int(max_det), ws.data_ptr(), ws.numel(), rows.data_ptr(), idx.data_ptr(), cnt.data_ptr(),
                                   (lib.NMS_FLOAT_THRESHOLD if rule == "cuda" else 0) | (lib.NMS_SINGLE_LAUNCH if _single_launch(B) else 0)
                                   | (lib.NMS_PRECOLLECTED if pre else 0) | (lib.NMS_MATRIX if _matrix_form(B, stream is not None) else 0), st.cuda_stream))
            if pre:
                ev = torch.cuda.Event()
                ev.record(st)
                ws._maf_busy = ev                    # the next forward that refills this workspace waits for it (candidate_workspace)
                prediction._maf_cand = None          # the lists are consumed: the sort rewrites the keys in place
        if stream is not None and pred is prediction:
            pred.record_stream(st)                    # a caller-owned tensor read on a stream it was not allocated on
    return rows, idx, cnt


SINGLE_LAUNCH_MAX_BATCH = cfg.nms_single_max_batch
MATRIX_PATH = {"0": False, "1": True}.get(str(cfg.nms_matrix), "auto")       # the all-pairs path as n x n suppression matrix + row scan (True), as kept-list scan (False), or by situation ("auto": _matrix_form)


def _single_launch(B):
    """One kernel for the whole NMS (csrc/nms.hip:nms_single_kernel) up to this batch size: the latency path.  Larger batches keep the
    seven-launch form, whose suppression-matrix kernels use the whole chip."""
    return B <= SINGLE_LAUNCH_MAX_BATCH


class NmsHandle:
    """In-flight NMS of one batch (see non_max_suppression_async)."""

    def __init__(self, rows, idx, cnt, event, keep=None):
        self.rows, self.idx, self.cnt, self.event = rows, idx, cnt, event
        self._keep = keep                                  # the prediction tensor: alive until the side stream has read it

    def result(self, return_index=False):
        self.event.synchronize()
        self._keep = None
        counts = self.cnt.tolist()
        out = [self.rows[b, :n] for b, n in enumerate(counts)]
        if return_index:
            return out, [self.idx[b, :n] for b, n in enumerate(counts)]
        return out


def non_max_suppression_async(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300, side=None, iou_rule=None):
    """Same arguments as non_max_suppression, but the kernels go to a side stream ordered after the current one and
    the call returns at once; `handle.result()` gives the reference's list of tensors.  A serving loop calls this for
    batch i, launches the forward of batch i+1, then collects batch i: the (latency-bound, 32-workgroup) NMS of one
    batch overlaps the convolutions of the next.  `side`: the stream to use (default: one per device, created on first use; streams
    that share a hardware queue do not overlap — see streams.concurrent_streams)."""
    dev = prediction.device
    if side is None:                                       # `side`: the caller's stream for the NMS (e.g. one of streams.concurrent_streams)
        side = _side.get(dev.index)
        if side is None:
            side = _side[dev.index] = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    prediction.record_stream(side)                         # the caching allocator must not hand this block out again before the NMS has read it
    rows, idx, cnt = nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, stream=side, iou_rule=iou_rule)
    ev = torch.cuda.Event()
    ev.record(side)
    return NmsHandle(rows, idx, cnt, ev, keep=prediction)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        max_det=300, return_index=False, iou_rule=None):
    rows, idx, cnt = nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, iou_rule=iou_rule)
    counts = cnt.tolist()                              # the only device->host sync
    out = [rows[b, :n] for b, n in enumerate(counts)]
    if return_index:
        return out, [idx[b, :n] for b, n in enumerate(counts)]
    return out
