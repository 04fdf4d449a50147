"""Golden vectors for the training loss: the reference's own ComputeLoss (yolov6/models/loss.py) with its
TaskAlignedAssigner (c* keys) and with its warm-up ATSSAssigner (a* keys), run here in the build container on seeded head outputs and labels.

    python tools/make_golden_loss.py    ->  tests/golden/loss_cases.npz

The reference moves two parameter-free sub-modules to the GPU in its constructor (`VarifocalLoss().cuda()`, loss.py:46-47); this
container has no GPU, so nn.Module.cuda is made a no-op for the run — everything then computes on the CPU in fp32."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402


def cases():
    out = []
    for ci, (B, size, nt) in enumerate([(2, 64, [3, 0]), (3, 128, [5, 9, 1]), (2, 96, [0, 0]), (1, 160, [14])]):
        g = torch.Generator().manual_seed(100 + ci)
        hw = [(size // s, size // s) for s in (8, 16, 32)]
        A = sum(h * w for h, w in hw)
        scores = torch.sigmoid(torch.randn(B, A, 80, generator=g) * 1.5 - 2.0)
        distri = torch.randn(B, A, 68, generator=g) * 1.2
        rows = []
        for b, n in enumerate(nt):
            for _ in range(n):
                cx, cy = torch.rand(2, generator=g).tolist()
                w, h = (torch.rand(2, generator=g) * 0.5 + 0.08).tolist()
                rows.append([b, int(torch.randint(0, 80, (1,), generator=g)), cx, cy, w, h])
        targets = torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)
        out.append((size, hw, scores, distri, targets))
    return out


def main():
    ref_import.load(lambda b, s, t: torch.zeros(0, dtype=torch.long))
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, ref_import.REF)
    from yolov6.models.loss import ComputeLoss
    blob = {}
    for ci, (size, hw, scores, distri, targets) in enumerate(cases()):
        crit = ComputeLoss(num_classes=80, ori_img_size=size, warmup_epoch=0, use_dfl=True, reg_max=16, iou_type="giou")
        feats = [torch.zeros(scores.shape[0], 8, h, w) for h, w in hw]
        s = scores.clone().requires_grad_(True); d = distri.clone().requires_grad_(True)
        loss, items = crit((feats, s, d), targets.clone(), 5, 1)
        if torch.isfinite(loss):
            loss.backward()
        blob["c%d_size" % ci] = np.asarray(size)
        blob["c%d_scores" % ci] = scores.numpy(); blob["c%d_distri" % ci] = distri.numpy(); blob["c%d_targets" % ci] = targets.numpy()
        blob["c%d_loss" % ci] = np.asarray(loss.item()); blob["c%d_items" % ci] = items.numpy()
        blob["c%d_gscores" % ci] = s.grad.numpy() if s.grad is not None else np.zeros(0)
        blob["c%d_gdistri" % ci] = d.grad.numpy() if d.grad is not None else np.zeros(0)
        print("case", ci, "loss", loss.item(), items.tolist())
        # the same inputs through the warm-up assigner: ComputeLoss's default warmup_epoch = 3 is what the trainer uses (engine.py:303-308), epoch 0
        if min(h * w for h, w in hw) < 9:
            continue                       # the reference's ATSS raises when a level has fewer than topk = 9 anchors (atss_assigner.py:104)
        crit = ComputeLoss(num_classes=80, ori_img_size=size, use_dfl=True, reg_max=16, iou_type="giou")
        s = scores.clone().requires_grad_(True); d = distri.clone().requires_grad_(True)
        loss, items = crit((feats, s, d), targets.clone(), 0, 1)
        if torch.isfinite(loss):
            loss.backward()
        blob["a%d_loss" % ci] = np.asarray(loss.item()); blob["a%d_items" % ci] = items.numpy()
        blob["a%d_gscores" % ci] = s.grad.numpy().astype(np.float16) if s.grad is not None else np.zeros(0)
        blob["a%d_gdistri" % ci] = d.grad.numpy() if d.grad is not None else np.zeros(0)
        print("case", ci, "ATSS loss", loss.item(), items.tolist())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_cases.npz"), **blob)


if __name__ == "__main__":
    main()
