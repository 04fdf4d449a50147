"""Deterministic synthetic `prediction` tensors for NMS edge-case tests (NumPy legacy RNG).

Used by tools/make_golden.py (to record what the reference's non_max_suppression returns for them)
and by the parity tests (to feed the same inputs to the oracle and to the HIP path).
Each case: name -> (prediction [B,N,5+nc] fp32, kwargs for non_max_suppression)."""
import numpy as np


def _boxes(rs, n, img=640.0):
    cxy = rs.rand(n, 2) * img
    wh = 8.0 + rs.rand(n, 2) * 200.0
    return np.concatenate([cxy, wh], 1).astype(np.float32)


def cases():
    out = {}
    nc = 80
    rs = np.random.RandomState(7)

    # 1. clustered boxes, sparse class scores, eval settings (multi-label, per-class offsets)
    n = 600
    base = _boxes(rs, 40)
    b = base[rs.randint(0, 40, n)] + rs.randn(n, 4).astype(np.float32) * 6.0
    cls = (rs.rand(n, nc) ** 12).astype(np.float32)
    p = np.concatenate([b, np.ones((n, 1), np.float32), cls], 1)[None]
    p2 = p.copy(); p2[0, :, 5:] = (rs.rand(n, nc) ** 10).astype(np.float32)
    out["clustered_eval"] = (np.concatenate([p, p2], 0), dict(conf_thres=0.03, iou_thres=0.65, multi_label=True))
    out["clustered_agnostic"] = (np.concatenate([p, p2], 0), dict(conf_thres=0.4, iou_thres=0.45, agnostic=True, max_det=1000))
    out["clustered_bestclass"] = (np.concatenate([p, p2], 0), dict(conf_thres=0.25, iou_thres=0.45))
    out["clustered_classes"] = (np.concatenate([p, p2], 0), dict(conf_thres=0.1, iou_thres=0.5, classes=[0, 3, 17, 79], multi_label=True))

    # 2. empty: nothing above conf; and one image empty / one not
    q = p.copy(); q[0, :, 5:] *= 0.01
    out["empty_all"] = (q, dict(conf_thres=0.25, iou_thres=0.45, multi_label=True))
    out["empty_mixed"] = (np.concatenate([q, p], 0), dict(conf_thres=0.25, iou_thres=0.45, multi_label=True))

    # 3. exact ties: duplicated boxes with identical scores (tie rule: lower candidate index first)
    t = p[:, :64].copy()
    t[0, 32:64] = t[0, 0:32]
    t[0, :, 5:] = 0.0
    t[0, :, 5 + 3] = 0.5
    t[0, 10:20, 5 + 3] = 0.75
    t[0, 42:52, 5 + 3] = 0.75
    out["ties"] = (t, dict(conf_thres=0.25, iou_thres=0.45, multi_label=True))

    # 4. thresholds hit exactly: score == conf_thres must be dropped (strict >), objectness != 1
    e = p[:, :128].copy()
    e[0, :, 4] = (0.5 + 0.5 * rs.rand(128)).astype(np.float32)
    e[0, :, 5:] = 0.0
    e[0, :, 5 + 7] = (rs.rand(128)).astype(np.float32)
    e[0, 5, 4] = 1.0; e[0, 5, 5 + 7] = np.float32(0.25)               # == thr -> dropped
    e[0, 6, 4] = 1.0; e[0, 6, 5 + 7] = np.nextafter(np.float32(0.25), np.float32(1))  # kept
    e[0, 7, 4] = np.float32(0.25); e[0, 7, 5 + 7] = 1.0               # obj == thr -> dropped
    out["strict_thresholds"] = (e, dict(conf_thres=0.25, iou_thres=0.45, multi_label=True))

    # 5. more than max_nms=30000 candidates -> top-30000 by score, then NMS; tiny boxes so many survive
    n = 520
    b = _boxes(rs, n); b[:, 2:] = 4.0 + rs.rand(n, 2).astype(np.float32) * 8.0
    cls = (0.05 + 0.9 * rs.rand(n, nc)).astype(np.float32)
    out["overflow_30000"] = (np.concatenate([b, np.ones((n, 1), np.float32), cls], 1)[None],
                             dict(conf_thres=0.03, iou_thres=0.65, multi_label=True))

    # 6. max_det truncation with many isolated boxes
    n = 900
    g = np.stack(np.meshgrid(np.arange(30), np.arange(30)), -1).reshape(-1, 2).astype(np.float32) * 20 + 10
    b = np.concatenate([g, np.full((n, 2), 10.0, np.float32)], 1)
    cls = np.zeros((n, nc), np.float32); cls[np.arange(n), rs.randint(0, nc, n)] = (0.3 + 0.7 * rs.rand(n)).astype(np.float32)
    out["max_det"] = (np.concatenate([b, np.ones((n, 1), np.float32), cls], 1)[None],
                      dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300))

    # 7. single class (multi_label is forced off when nc == 1, nms.py:57)
    n = 200
    b = base[rs.randint(0, 40, n)] + rs.randn(n, 4).astype(np.float32) * 5.0
    out["single_class"] = (np.concatenate([b, rs.rand(n, 1).astype(np.float32), rs.rand(n, 1).astype(np.float32)], 1)[None].astype(np.float32),
                           dict(conf_thres=0.2, iou_thres=0.5, multi_label=True))

    # 8. boxes that leave their 4096-px class band: huge widths, negative corners, far-away centres
    #    (cross-class overlaps through the offset trick, nms.py:94-95) + a few degenerate (w <= 0) boxes
    n = 300
    b = _boxes(rs, n)
    b[:60, 2] = 3000.0 + rs.rand(60).astype(np.float32) * 9000.0          # very wide: spans 2-4 bands
    b[60:90, 0] = -2000.0 + rs.rand(30).astype(np.float32) * 1000.0        # far left of the image
    b[90:120, 0] = 4000.0 + rs.rand(30).astype(np.float32) * 400.0         # straddles the next class band
    b[120:130, 2] = -5.0                                                   # negative width
    cls = np.zeros((n, nc), np.float32)
    cls[np.arange(n), rs.randint(0, 6, n)] = (0.3 + 0.7 * rs.rand(n)).astype(np.float32)
    cls[np.arange(n), rs.randint(0, 6, n)] = (0.3 + 0.7 * rs.rand(n)).astype(np.float32)
    out["cross_band"] = (np.concatenate([b, np.ones((n, 1), np.float32), cls], 1)[None],
                         dict(conf_thres=0.25, iou_thres=0.3, multi_label=True))
    return out


# ---- shared by the detection-level parity tests (GPU and CPU): one-to-one matching of detection rows
def _iou(a, b):
    """a [n,4], b [m,4] xyxy -> [n,m]"""
    lt = np.maximum(a[:, None, :2], b[None, :, :2]); rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None] - inter + 1e-12)


def match_detections(got, ref, iou_min=0.95, dscore=1e-2):
    """Greedy one-to-one matching of detection rows (x1,y1,x2,y2,conf,cls) by class, IoU >= iou_min and |d score| <= dscore.
    Returns (matched pairs, unmatched reference rows, unmatched rows of `got`)."""
    iou = _iou(ref[:, :4], got[:, :4])
    ok = (iou >= iou_min) & (ref[:, None, 5] == got[None, :, 5]) & (np.abs(ref[:, None, 4] - got[None, :, 4]) <= dscore)
    used = np.zeros(got.shape[0], bool)
    pairs, miss = [], []
    for i in range(ref.shape[0]):
        cand = np.where(ok[i] & ~used)[0]
        if cand.size == 0:
            miss.append(i)
            continue
        j = cand[np.argmax(iou[i, cand])]
        used[j] = True
        pairs.append((i, j))
    return pairs, miss, np.where(~used)[0].tolist()
