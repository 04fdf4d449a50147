"""Training loss on the device without host round trips.

`ComputeLoss` has the constructor arguments, the call signature `(outputs, targets, epoch_num, step_num)` and the return value
`(loss, [iou, dfl, cls] items)` of the reference's ComputeLoss (yolov6/models/loss.py:15-193): VariFocal classification loss, GIoU
box loss and Distribution Focal Loss over the targets of the task-aligned assigner.  What differs is how it gets there:

* labels stay on the device: the reference pads them per image through Python lists and `targets.cpu().numpy()` (loss.py:179-188, a
  host sync every step); here one small kernel groups them by image and converts the boxes;
* the assignment is one HIP kernel on the ragged boxes (csrc/tal_assign.hip) instead of dense [B, n_max, 8400] masks, `one_hot` of the
  top-k indices and the try/except that falls back to the CPU when those masks run out of memory (loss.py:82-149);
* the loss terms are three more kernels (csrc/loss_terms.hip): the box decode that feeds the assigner, and VariFocal / GIoU + DFL sums
  with their gradients straight from the assigner's two per-anchor arrays — the [B,A,nc] one-hot labels and score targets of the
  reference are never built.  `fused=False` keeps the terms as torch ops on the device (the A/B the kernels are tested against).

Both assigners of the reference are here: ATSS for `epoch_num < warmup_epoch` (loss.py:83-91; csrc/tal_assign.hip atss_cand_kernel: the
9 nearest anchors per level from a 9 x 9 window around the box centre instead of a distance matrix over all anchors) and the
task-aligned assigner afterwards; they share the label preprocessing, the resolve kernel and the loss kernels.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import lib


def _anchors(feats, strides, offset, device):
    """anchor_generator.py:26-53: centres in pixels [A,2] and the stride of every anchor [A,1]."""
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[-2:]
        ys = (torch.arange(h, device=device, dtype=torch.float32) + offset) * s
        xs = (torch.arange(w, device=device, dtype=torch.float32) + offset) * s
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pts.append(torch.stack([xx, yy], -1).reshape(-1, 2))
        st.append(torch.full((h * w, 1), float(s), device=device))
    return torch.cat(pts).contiguous(), torch.cat(st)


def _targets_on_device(targets, batch_size, img_size, dev):
    """loss.py:179-188 without the host (csrc/tal_assign.hip tal_targets_kernel): rows grouped by image, boxes as pixel xyxy, offsets."""
    t = targets.to(dev, torch.float32).contiguous()
    T = t.shape[0]
    gts = (torch.empty if T else torch.zeros)(max(T, 1), 5, dtype=torch.float32, device=dev)     # T = 0: one all-zero row nobody is assigned to
    gt_img = torch.empty(max(T, 1), dtype=torch.int32, device=dev)
    offs = torch.empty(batch_size + 1, dtype=torch.int32, device=dev)
    lib.check(lib.load().maf_tal_targets(t.data_ptr(), T, batch_size, float(img_size), gts.data_ptr(), gt_img.data_ptr(), offs.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream))
    return gts, gt_img, offs, T


def _levels(level_hw, strides, offset):
    """The anchor grids as the host arrays maf_tal_assign takes: (n, int32 [n][2], float [n], cell offset)."""
    n = len(level_hw)
    hw = (ctypes.c_int32 * (2 * n))(*[int(v) for p in level_hw for v in p])
    st = (ctypes.c_float * n)(*[float(v) for v in strides])
    return n, hw, st, float(offset)


def _assign(pred_scores, pred_bboxes, anchor_points, levels, gts, gt_img, offs, T, topk, alpha, beta):
    """csrc/tal_assign.hip -> (row of the assigned box or -1 [B,A] int32, normalised alignment metric [B,A] fp32)."""
    dev = pred_scores.device
    B, A, nc = pred_scores.shape
    ps = pred_scores.detach()
    if ps.dtype not in (torch.float16, torch.float32):
        ps = ps.float()
    ps = ps.contiguous()
    pb = pred_bboxes.detach().float().contiguous()
    out_gt = torch.empty(B, A, dtype=torch.int32, device=dev)
    out_norm = torch.empty(B, A, dtype=torch.float32, device=dev)
    cand = torch.empty(max(T, 1) * topk, dtype=torch.int32, device=dev)
    lib.check(lib.load().maf_tal_assign(ps.data_ptr(), lib.F16 if ps.dtype == torch.float16 else lib.F32, pb.data_ptr(), anchor_points.data_ptr(),
                                        gts.data_ptr(), gt_img.data_ptr(), offs.data_ptr(), T, B, A, nc, topk, float(alpha), float(beta), 1e-9,
                                        levels[0], ctypes.cast(levels[1], ctypes.c_void_p), ctypes.cast(levels[2], ctypes.c_void_p), levels[3], cand.data_ptr(), out_gt.data_ptr(), out_norm.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return out_gt, out_norm


def _assign_atss(pred_bboxes, anchor_points, levels, gts, gt_img, offs, T, topk=9, cell_size=5.0):
    """csrc/tal_assign.hip atss_cand_kernel + resolve -> (row of the assigned box or -1, IoU of the predicted box with it)."""
    dev = pred_bboxes.device
    B, A = pred_bboxes.shape[:2]
    pb = pred_bboxes.detach().float().contiguous()
    out_gt = torch.empty(B, A, dtype=torch.int32, device=dev)
    out_norm = torch.empty(B, A, dtype=torch.float32, device=dev)
    cand = torch.empty(max(T, 1) * topk * levels[0], dtype=torch.int32, device=dev)
    lib.check(lib.load().maf_atss_assign(pb.data_ptr(), anchor_points.data_ptr(), gts.data_ptr(), gt_img.data_ptr(), offs.data_ptr(), T, B, A, topk,
                                         levels[0], ctypes.cast(levels[1], ctypes.c_void_p), ctypes.cast(levels[2], ctypes.c_void_p), levels[3],
                                         float(cell_size), cand.data_ptr(), out_gt.data_ptr(), out_norm.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream))
    return out_gt, out_norm


def task_aligned_assign(pred_scores, pred_bboxes, anchor_points, targets, batch_size, img_size, num_classes=80, topk=13, alpha=1.0, beta=6.0,
                        level_hw=None, strides=(8, 16, 32), cell_offset=0.5):
    """level_hw: (rows, columns) of every anchor grid (default: img_size / stride squares); targets [T,6] = (image, class, cx, cy, w, h) normalised -> (target_labels [B,A] long, target_bboxes [B,A,4] pixels,
    target_scores [B,A,nc], fg_mask [B,A] bool) as TaskAlignedAssigner.forward returns them, all on the device, no synchronisation.
    (The fused loss never builds these; this is the reference-shaped view of the assigner's output.)"""
    dev = pred_scores.device
    gts, gt_img, offs, T = _targets_on_device(targets, batch_size, img_size, dev)
    level_hw = level_hw or [(int(img_size) // s, int(img_size) // s) for s in strides]
    out_gt, out_norm = _assign(pred_scores, pred_bboxes, anchor_points, _levels(level_hw, strides, cell_offset), gts, gt_img, offs, T, topk, alpha, beta)
    fg = out_gt >= 0
    idx = out_gt.clamp(min=0).long()
    labels = gts[:, 0].long()[idx]
    boxes = gts[:, 1:][idx] * fg.unsqueeze(-1)
    scores = F.one_hot(labels, pred_scores.shape[-1]).float() * (out_norm * fg).unsqueeze(-1)
    return labels, boxes, scores, fg


class _FusedTerms(torch.autograd.Function):
    """out[5] = (weighted total, w_iou*iou, w_dfl*dfl, w_cls*cls, target-score sum) from csrc/loss_terms.hip; backward re-runs the kernels
    in their gradient form with the upstream gradient read from device memory, and writes the gradients in the dtype of the head outputs."""

    @staticmethod
    def forward(ctx, scores, distri, pts, st, gts, out_gt, out_norm, reg_max, weights):
        B, A, nc = scores.shape
        L = lib.load()
        dt = lib.F16 if scores.dtype == torch.float16 else lib.F32
        part = torch.empty(L.maf_loss_partial_rows(B, A, nc), 4, dtype=torch.float32, device=scores.device)
        out = torch.empty(5, dtype=torch.float32, device=scores.device)
        lib.check(L.maf_loss_terms(scores.data_ptr(), distri.data_ptr(), dt, pts.data_ptr(), st.data_ptr(), gts.data_ptr(), out_gt.data_ptr(),
                                   out_norm.data_ptr(), B, A, nc, reg_max, weights[0], weights[1], weights[2], part.data_ptr(), out.data_ptr(),
                                   None, None, None, torch.cuda.current_stream(scores.device).cuda_stream))
        ctx.save_for_backward(scores, distri, pts, st, gts, out_gt, out_norm, out)
        ctx.reg_max, ctx.weights = reg_max, weights
        return out

    @staticmethod
    def backward(ctx, g):
        scores, distri, pts, st, gts, out_gt, out_norm, out = ctx.saved_tensors
        B, A, nc = scores.shape
        L = lib.load()
        dt = lib.F16 if scores.dtype == torch.float16 else lib.F32
        up = g[:1].float().contiguous()                      # only the total carries gradient: the items are reported detached (loss.py:173-176)
        gs, gd = torch.empty_like(scores), torch.empty_like(distri)
        w = ctx.weights
        lib.check(L.maf_loss_terms(scores.data_ptr(), distri.data_ptr(), dt, pts.data_ptr(), st.data_ptr(), gts.data_ptr(), out_gt.data_ptr(),
                                   out_norm.data_ptr(), B, A, nc, ctx.reg_max, w[0], w[1], w[2], None, out.data_ptr(), up.data_ptr(),
                                   gs.data_ptr(), gd.data_ptr(), torch.cuda.current_stream(scores.device).cuda_stream))
        return gs, gd, None, None, None, None, None, None, None


def _giou_loss(b1, b2, eps=1e-10):
    """figure_iou.py IOUloss(box_format='xyxy', iou_type='giou', eps=1e-10)."""
    x1, y1, x2, y2 = b1.unbind(-1); u1, v1, u2, v2 = b2.unbind(-1)
    inter = (torch.min(x2, u2) - torch.max(x1, u1)).clamp(0) * (torch.min(y2, v2) - torch.max(y1, v1)).clamp(0)
    union = (x2 - x1) * (y2 - y1 + eps) + (u2 - u1) * (v2 - v1 + eps) - inter + eps
    iou = inter / union
    c_area = (torch.max(x2, u2) - torch.min(x1, u1)) * (torch.max(y2, v2) - torch.min(y1, v1)) + eps
    return 1.0 - (iou - (c_area - union) / c_area)


class ComputeLoss:
    def __init__(self, fpn_strides=(8, 16, 32), grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80, ori_img_size=640, warmup_epoch=3,
                 use_dfl=True, reg_max=16, iou_type="giou", loss_weight=None, fused=True):
        # warmup_epoch = 3 is the reference's default (loss.py:23) and its trainer does not override it (engine.py:303-308): ATSS assigns the
        # labels in epochs 0-2, the task-aligned assigner afterwards.  Pass 0 for the task-aligned assigner from the first step.
        assert use_dfl and iou_type == "giou", "MAF-YOLO trains with DFL + GIoU (configs/MAF-YOLO-n.py:14-16)"
        self.fpn_strides, self.grid_cell_offset, self.grid_cell_size, self.warmup_epoch = tuple(fpn_strides), grid_cell_offset, grid_cell_size, warmup_epoch
        self.num_classes, self.ori_img_size, self.reg_max = num_classes, ori_img_size, reg_max
        self.loss_weight = loss_weight or {"class": 1.0, "iou": 2.5, "dfl": 0.5}
        self.topk, self.alpha, self.beta = 13, 1.0, 6.0                       # loss.py:46
        self.fused = fused                     # False: same assigner kernels, loss terms as torch ops (the A/B the fused kernels are tested against)
        self._cache = {}
        self.last_assignment = None

    def __call__(self, outputs, targets, epoch_num=0, step_num=0, assignment=None):
        """(loss, items) as the reference's ComputeLoss.__call__ (loss.py:51-176).  `assignment` (not in the reference): a label assignment to use
        instead of running the assigner — the `last_assignment` = (box row per anchor or -1 [B,A] int32, normalised metric [B,A] fp32) of an
        earlier call on the same labels; parity tests freeze the discrete top-k choices of an fp32 pass this way when they run the fp16 pass."""
        feats, pred_scores, pred_distri = outputs
        dev = pred_scores.device
        if not pred_scores.is_cuda:
            raise lib.MafError("ComputeLoss runs on the HIP path only: got %s tensors (no CPU fallback)" % dev)
        B, A, nc = pred_scores.shape
        key = (tuple(tuple(f.shape[-2:]) for f in feats), dev.index)
        if key not in self._cache:
            pts, st = _anchors(feats, self.fpn_strides, self.grid_cell_offset, dev)
            self._cache = {key: (pts, st, st.reshape(-1).contiguous(), _levels(key[0], self.fpn_strides, self.grid_cell_offset))}
        pts, st, st_flat, levels = self._cache[key]
        gts, gt_img, offs, T = _targets_on_device(targets, B, self.ori_img_size, dev)
        lw = self.loss_weight
        if self.fused:
            if pred_scores.dtype != pred_distri.dtype or pred_scores.dtype not in (torch.float16, torch.float32):
                pred_scores, pred_distri = pred_scores.float(), pred_distri.float()
            ps, pd = pred_scores.contiguous(), pred_distri.contiguous()
            boxes = torch.empty(B, A, 4, dtype=torch.float32, device=dev)
            lib.check(lib.load().maf_loss_decode(pd.data_ptr(), lib.F16 if pd.dtype == torch.float16 else lib.F32, pts.data_ptr(), st_flat.data_ptr(),
                                                 B, A, self.reg_max, boxes.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            if assignment is not None:
                out_gt, out_norm = assignment
            elif epoch_num < self.warmup_epoch:                                 # loss.py:83-91
                out_gt, out_norm = _assign_atss(boxes, pts, levels, gts, gt_img, offs, T, 9, self.grid_cell_size)
            else:
                out_gt, out_norm = _assign(ps, boxes, pts, levels, gts, gt_img, offs, T, self.topk, self.alpha, self.beta)
            self.last_assignment = (out_gt, out_norm)
            out = _FusedTerms.apply(ps, pd, pts, st_flat, gts, out_gt, out_norm, self.reg_max, (float(lw["class"]), float(lw["iou"]), float(lw["dfl"])))
            return out[0], out[1:4].detach()
        loss_cls, loss_iou, loss_dfl = self._torch_terms(pred_scores, pred_distri, pts, st, levels, gts, gt_img, offs, T, epoch_num < self.warmup_epoch)
        loss = lw["class"] * loss_cls + lw["iou"] * loss_iou + lw["dfl"] * loss_dfl
        items = torch.stack([lw["iou"] * loss_iou, lw["dfl"] * loss_dfl, lw["class"] * loss_cls]).detach()
        return loss, items

    def _torch_terms(self, pred_scores, pred_distri, pts, st, levels, gts, gt_img, offs, T, warmup):
        dev = pred_scores.device
        B, A, nc = pred_scores.shape
        pts_s = pts / st
        R = self.reg_max
        proj = torch.linspace(0, R, R + 1, device=dev)
        pd = pred_distri.float().view(B, A, 4, R + 1)
        dist = F.softmax(pd, -1).matmul(proj)                                   # loss.py:190-193
        pred_bboxes = torch.cat([pts_s - dist[..., :2], pts_s + dist[..., 2:]], -1)
        ps = pred_scores.float()
        if warmup:
            out_gt, out_norm = _assign_atss(pred_bboxes * st, pts, levels, gts, gt_img, offs, T, 9, self.grid_cell_size)
        else:
            out_gt, out_norm = _assign(ps, pred_bboxes * st, pts, levels, gts, gt_img, offs, T, self.topk, self.alpha, self.beta)
        fg = out_gt >= 0
        idx = out_gt.clamp(min=0).long()
        labels = gts[:, 0].long()[idx]
        t_boxes = gts[:, 1:][idx] * fg.unsqueeze(-1) / st                       # loss.py:152
        t_scores = F.one_hot(labels, nc).float() * (out_norm * fg).unsqueeze(-1)
        # VariFocal loss (loss.py:196-206)
        one_hot = F.one_hot(torch.where(fg, labels, torch.full_like(labels, nc)), nc + 1)[..., :-1].float()
        w = 0.75 * ps.pow(2.0) * (1 - one_hot) + t_scores * one_hot             # the reference lets the gradient flow through the weight too
        tss = t_scores.sum()
        loss_cls = (F.binary_cross_entropy(ps, t_scores, reduction="none") * w).sum() / tss
        # box losses over the foreground anchors (loss.py:217-267), masked instead of gathered: no data-dependent shapes, no sync
        bw = t_scores.sum(-1) * fg
        zero = torch.zeros((), device=dev)
        loss_iou = torch.where(tss > 0, (_giou_loss(pred_bboxes, t_boxes) * bw).sum() / tss, zero)     # no foreground: zeros (loss.py:262-266)
        ltrb = torch.cat([pts_s - t_boxes[..., :2], t_boxes[..., 2:] - pts_s], -1).clip(0, R - 0.01)
        tl = ltrb.long()
        wl = (tl + 1).float() - ltrb
        logp = F.log_softmax(pd, -1)
        ce_l = -logp.gather(-1, tl.unsqueeze(-1)).squeeze(-1)
        ce_r = -logp.gather(-1, (tl + 1).unsqueeze(-1)).squeeze(-1)
        loss_dfl = torch.where(tss > 0, ((ce_l * wl + ce_r * (1 - wl)).mean(-1) * bw).sum() / tss, zero)
        return loss_cls, loss_iou, loss_dfl
