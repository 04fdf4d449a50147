"""Training loss on the device without host round trips.

`ComputeLoss` has the constructor arguments, the call signature `(outputs, targets, epoch_num, step_num)` and the return value
`(loss, [iou, dfl, cls] items)` of the reference's ComputeLoss (yolov6/models/loss.py:15-193): VariFocal classification loss, GIoU
box loss and Distribution Focal Loss over the targets of the task-aligned assigner.  What differs is how it gets there:

* labels stay on the device: the reference pads them per image through Python lists and `targets.cpu().numpy()` (loss.py:179-188, a
  host sync every step); here they are sorted by image and bucketed with `bincount` / `cumsum`;
* the assignment is one HIP kernel on the ragged boxes (csrc/tal_assign.hip) instead of dense [B, n_max, 8400] masks, `one_hot` of the
  top-k indices and the try/except that falls back to the CPU when those masks run out of memory (loss.py:82-149);
* the loss terms themselves are differentiable torch ops on the device (autograd supplies the gradient into both head outputs).

The reference uses ATSS for the first `warmup_epoch` epochs (loss.py:83-91); this class uses the task-aligned assigner from the first
step (pass warmup_epoch=0 to the reference to compare) — ATSS is the remaining piece of this row.
"""
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


def task_aligned_assign(pred_scores, pred_bboxes, anchor_points, targets, batch_size, img_size, num_classes=80, topk=13, alpha=1.0, beta=6.0):
    """targets [T,6] = (image, class, cx, cy, w, h) normalised -> (target_labels [B,A] long, target_bboxes [B,A,4] pixels,
    target_scores [B,A,nc], fg_mask [B,A] bool), all on the device, no synchronisation."""
    dev = pred_scores.device
    B, A, nc = pred_scores.shape
    t = targets.to(dev, torch.float32)
    order = torch.sort(t[:, 0], stable=True)[1] if t.shape[0] else torch.zeros(0, dtype=torch.long, device=dev)
    t = t[order]
    xywh = t[:, 2:6] * float(img_size)
    gts = torch.cat([t[:, 1:2], xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2], 1).contiguous()       # loss.py:186-187
    counts = torch.bincount(t[:, 0].long(), minlength=batch_size)[:batch_size]
    offs = torch.zeros(batch_size + 1, dtype=torch.int32, device=dev)
    offs[1:] = torch.cumsum(counts, 0).int()
    if gts.shape[0] == 0:
        gts = torch.zeros(1, 5, device=dev)                                    # never read (all counts are 0); keeps the pointer valid
    out_gt = torch.empty(B, A, dtype=torch.int32, device=dev)
    out_norm = torch.empty(B, A, dtype=torch.float32, device=dev)
    ps = pred_scores.detach().float().contiguous()
    pb = pred_bboxes.detach().float().contiguous()
    lib.check(lib.load().maf_tal_assign(ps.data_ptr(), pb.data_ptr(), anchor_points.data_ptr(), gts.data_ptr(), offs.data_ptr(), B, A, nc, topk,
                                        float(alpha), float(beta), 1e-9, out_gt.data_ptr(), out_norm.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream))
    fg = out_gt >= 0
    idx = out_gt.clamp(min=0).long()
    labels = gts[:, 0].long()[idx]
    boxes = gts[:, 1:][idx] * fg.unsqueeze(-1)
    scores = F.one_hot(labels, nc).float() * (out_norm * fg).unsqueeze(-1)
    return labels, boxes, scores, fg


def _giou_loss(b1, b2, eps=1e-10):
    """figure_iou.py IOUloss(box_format='xyxy', iou_type='giou', eps=1e-10)."""
    x1, y1, x2, y2 = b1.unbind(-1); u1, v1, u2, v2 = b2.unbind(-1)
    inter = (torch.min(x2, u2) - torch.max(x1, u1)).clamp(0) * (torch.min(y2, v2) - torch.max(y1, v1)).clamp(0)
    union = (x2 - x1) * (y2 - y1 + eps) + (u2 - u1) * (v2 - v1 + eps) - inter + eps
    iou = inter / union
    c_area = (torch.max(x2, u2) - torch.min(x1, u1)) * (torch.max(y2, v2) - torch.min(y1, v1)) + eps
    return 1.0 - (iou - (c_area - union) / c_area)


class ComputeLoss:
    def __init__(self, fpn_strides=(8, 16, 32), grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80, ori_img_size=640, warmup_epoch=0,
                 use_dfl=True, reg_max=16, iou_type="giou", loss_weight=None):
        assert use_dfl and iou_type == "giou", "MAF-YOLO trains with DFL + GIoU (configs/MAF-YOLO-n.py:14-16)"
        self.fpn_strides, self.grid_cell_offset = tuple(fpn_strides), grid_cell_offset
        self.num_classes, self.ori_img_size, self.reg_max = num_classes, ori_img_size, reg_max
        self.loss_weight = loss_weight or {"class": 1.0, "iou": 2.5, "dfl": 0.5}
        self.topk, self.alpha, self.beta = 13, 1.0, 6.0                       # loss.py:46
        self._cache = {}

    def __call__(self, outputs, targets, epoch_num=0, step_num=0):
        feats, pred_scores, pred_distri = outputs
        dev = pred_scores.device
        if not pred_scores.is_cuda:
            raise lib.MafError("ComputeLoss runs on the HIP path only: got %s tensors (no CPU fallback)" % dev)
        B, A, nc = pred_scores.shape
        key = (tuple(tuple(f.shape[-2:]) for f in feats), dev.index)
        if key not in self._cache:
            self._cache = {key: _anchors(feats, self.fpn_strides, self.grid_cell_offset, dev)}
        pts, st = self._cache[key]
        pts_s = pts / st
        R = self.reg_max
        proj = torch.linspace(0, R, R + 1, device=dev)
        pd = pred_distri.float().view(B, A, 4, R + 1)
        dist = F.softmax(pd, -1).matmul(proj)                                   # loss.py:190-193
        pred_bboxes = torch.cat([pts_s - dist[..., :2], pts_s + dist[..., 2:]], -1)
        ps = pred_scores.float()
        labels, t_boxes, t_scores, fg = task_aligned_assign(ps, pred_bboxes * st, pts, targets, B, self.ori_img_size, nc, self.topk, self.alpha, self.beta)
        t_boxes = t_boxes / st                                                  # loss.py:152
        # VariFocal loss (loss.py:196-206); the one-hot of the assigned label is t_scores > 0 up to anchors whose norm is exactly 0
        one_hot = F.one_hot(torch.where(fg, labels, torch.full_like(labels, nc)), nc + 1)[..., :-1].float()
        w = 0.75 * ps.pow(2.0) * (1 - one_hot) + t_scores * one_hot             # the reference lets the gradient flow through the weight too
        loss_cls = (F.binary_cross_entropy(ps, t_scores, reduction="none") * w).sum()
        tss = t_scores.sum()
        loss_cls = loss_cls / tss
        # box losses over the foreground anchors (loss.py:217-267), masked instead of gathered: no data-dependent shapes, no sync
        bw = t_scores.sum(-1) * fg
        loss_iou = (_giou_loss(pred_bboxes, t_boxes) * bw).sum() / tss
        ltrb = torch.cat([pts_s - t_boxes[..., :2], t_boxes[..., 2:] - pts_s], -1).clip(0, R - 0.01)
        tl = ltrb.long()
        wl = (tl + 1).float() - ltrb
        logp = F.log_softmax(pd, -1)
        ce_l = -logp.gather(-1, tl.unsqueeze(-1)).squeeze(-1)
        ce_r = -logp.gather(-1, (tl + 1).unsqueeze(-1)).squeeze(-1)
        loss_dfl = ((ce_l * wl + ce_r * (1 - wl)).mean(-1) * bw).sum() / tss
        lw = self.loss_weight
        loss = lw["class"] * loss_cls + lw["iou"] * loss_iou + lw["dfl"] * loss_dfl
        items = torch.stack([lw["iou"] * loss_iou, lw["dfl"] * loss_dfl, lw["class"] * loss_cls]).detach()
        return loss, items
