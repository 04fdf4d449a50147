#!/usr/bin/env python3
"""Golden vectors for the TRAIN-FORM graph: the reference's own Model in train mode + its own ComputeLoss + autograd, run here
in the build container (CPU, fp32) on seeded weights, images and labels — what Trainer.train_in_steps computes up to `backward()`
(yolov6/core/engine.py:149-164, without autocast: there is no GPU here).

    python tools/make_golden_train.py [n|s|m]    ->  tests/golden/train_<scale>.npz     (BASELINE configs[2] / [3] train the s and m graphs)
    python tools/make_golden_train.py n 640      ->  tests/golden/train_n_640.npz       (the same step at the BASELINE image size: a full-resolution gradient, 2 x 640 x 640)

Stored (DATA only): the reference's own label ASSIGNMENT of each pass (what its assigner returned: foreground mask, assigned box, class and
target score per anchor — the parity test freezes both of its legs to it, so that no near-tied discrete choice can differ), the loss and its items, the train-branch head outputs on a strided set of anchors, the gradient of a spread of parameters
(every kind of layer: RepVGG 3x3 / 1x1 branches, ConvWrapper 3x3, 1x1 convs, every depth-wise kernel size, BatchNorm affine, head preds) as
checksums + a strided sample each, and the BatchNorm running statistics after the step.  Inputs are regenerated from seeds by the test.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import                       # noqa: E402
from oracle import maf_oracle as O      # noqa: E402

SIZE, BATCH = 128, 2


def inputs():
    x = O.synth_images(BATCH, SIZE, seed=7)
    targets = torch.tensor([[0, 3, 0.40, 0.50, 0.30, 0.40], [0, 17, 0.70, 0.30, 0.20, 0.50], [0, 17, 0.25, 0.75, 0.30, 0.25],
                            [1, 5, 0.50, 0.50, 0.60, 0.60], [1, 62, 0.20, 0.30, 0.25, 0.35]], dtype=torch.float32)
    if SIZE != 128:
        # at 640 every centre above is a multiple of 32 px — a corner of four cells on ALL three levels: the assigners' "k nearest anchors" (atss_assigner.py:90-110,
        # torch.topk on exactly tied distances) would then pin torch's unspecified tie order, not the assigner.  Centres moved off the grid by a few pixels.
        targets[:, 2] += torch.tensor([0.0047, -0.0061, 0.0033, 0.0071, -0.0029])
        targets[:, 3] += torch.tensor([-0.0053, 0.0037, 0.0059, -0.0043, 0.0067])
    return x, targets


def picked(names):
    """A spread of parameters: every layer kind, every stage."""
    want = []
    for frag in ("backbone.0.rbr_dense.conv.weight", "backbone.0.rbr_1x1.conv.weight", "backbone.0.rbr_dense.bn.weight", "backbone.1.rbr_dense.conv.weight",
                 "backbone.1.rbr_1x1.bn.bias", "backbone.2.conv1.conv.weight", "backbone.2.m.0.conv1.conv.weight", "backbone.2.m.0.conv2.dwconv.lk_origin.weight",
                 "backbone.2.m.0.conv2.dwconv.dil_conv_k3_1.weight", "backbone.2.m.0.conv2.norm.weight", "backbone.2.m.0.one_conv.conv.weight",
                 "backbone.3.conv1.conv.weight", "backbone.3.conv2.rbr_dense.conv.weight", "backbone.4.m.0.conv2.dwconv.lk_origin.weight",
                 "backbone.6.m.0.conv2.dwconv.lk_origin.weight", "backbone.8.m.0.conv2.dwconv.lk_origin.weight", "backbone.8.conv2.conv.weight",
                 "backbone.9.cv2.conv.weight", "backbone.10.block.conv.weight", "backbone.12.conv1.conv.weight", "backbone.16.conv1.conv.weight",
                 "backbone.18.block.bn.weight", "backbone.23.block.conv.weight", "backbone.26.conv2.conv.weight", "backbone.30.conv2.bn.bias",
                 "backbone.31.stem.conv.weight", "backbone.31.cls_conv.dwconv.lk_origin.weight", "backbone.31.cls_pred.weight", "backbone.31.reg_pred.bias",
                 "backbone.32.reg_conv_s.conv.weight", "backbone.33.cls_pred.bias", "backbone.33.reg_pred.weight"):
        assert frag in names, frag
        want.append(frag)
    # deeper scales (s: RepHDW depth 2, m: up to 4): the LAST bottleneck of a few stages as well
    for frag in ("backbone.4.m.1.conv2.dwconv.lk_origin.weight", "backbone.6.m.3.one_conv.conv.weight", "backbone.12.m.2.conv1.conv.weight", "backbone.22.m.1.conv2.norm.bias",
                 "backbone.26.m.2.one_conv.conv.weight"):
        if frag in names:
            want.append(frag)
    return want


def summary(t):
    a = t.detach().double().reshape(-1).numpy()
    ramp = (np.arange(a.size) % 97 + 1).astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * ramp).sum(), np.abs(a).max()])


def main():
    global SIZE
    scale = sys.argv[1] if len(sys.argv) > 1 else "n"
    assert scale in ("n", "s", "m")
    if len(sys.argv) > 2:
        SIZE = int(sys.argv[2])
    suffix = "" if SIZE == 128 else "_%d" % SIZE
    torch.set_num_threads(os.cpu_count())
    ns = ref_import.load(lambda b, s, t: torch.zeros(0, dtype=torch.long))
    torch.nn.Module.cuda = lambda self, *a, **k: self          # ComputeLoss moves parameter-free sub-modules to the GPU in its constructor (loss.py:46-47)
    sys.path.insert(0, ref_import.REF)
    from yolov6.models.loss import ComputeLoss
    model = ref_import.build(ns, scale)
    model.load_state_dict(O.synth_state_dict(scale, seed=0), strict=True)
    model.train()
    x, targets = inputs()
    blob = {}
    for tag, epoch, kw in (("tal", 5, dict(warmup_epoch=0)), ("atss", 0, dict())):      # the steady-state assigner and the trainer's warm-up default (loss.py:23)
        model.load_state_dict(O.synth_state_dict(scale, seed=0), strict=True)
        model.zero_grad(set_to_none=True)
        crit = ComputeLoss(num_classes=80, ori_img_size=SIZE, use_dfl=True, reg_max=16, iou_type="giou", **kw)
        seen = {}

        def grab(mod, args, out):                               # (target_labels, target_bboxes px, target_scores, fg_mask): loss.py:83-100
            seen["a"] = [t.detach().clone() for t in out]
        hooks = [crit.warmup_assigner.register_forward_hook(grab), crit.formal_assigner.register_forward_hook(grab)]
        preds, _ = model(x)                                     # engine.py:150
        loss, items = crit(preds, targets.clone(), epoch, 1)    # engine.py:160
        loss.backward()                                         # engine.py:164 (no GradScaler on the CPU)
        for h in hooks:
            h.remove()
        t_lab, t_box, t_sc, fg = seen["a"]
        blob[tag + "_asg_fg"] = (fg > 0).numpy().astype(np.uint8)
        blob[tag + "_asg_box"] = t_box.float().numpy()
        blob[tag + "_asg_label"] = t_lab.long().numpy().astype(np.int32)
        blob[tag + "_asg_score"] = t_sc.float().sum(-1).numpy()   # one_hot(label) * normalised metric: the sum is the metric
        feats, cls, reg = preds
        blob[tag + "_loss"] = np.asarray(loss.item()); blob[tag + "_items"] = items.numpy()
        blob[tag + "_cls_rows"] = cls[:, ::37].detach().numpy(); blob[tag + "_reg_rows"] = reg[:, ::37].detach().numpy()
        params = dict(model.named_parameters())
        names = picked(set(params))
        blob["names"] = np.array(names)
        for i, n in enumerate(names):
            gr = params[n].grad
            blob["%s_g%d_sum" % (tag, i)] = summary(gr)
            blob["%s_g%d_sample" % (tag, i)] = gr.reshape(-1)[::max(1, gr.numel() // 64)][:64].numpy()
        print(tag, "loss", loss.item(), items.tolist(), "grad max", max(float(summary(params[n].grad)[3]) for n in names))
    # BatchNorm running statistics after ONE train-mode forward (momentum 0.03, torch_utils.py:43-45)
    model.load_state_dict(O.synth_state_dict(scale, seed=0), strict=True)
    with torch.no_grad():
        model(x)
    sd = model.state_dict()
    bn_names = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    pick = bn_names[::max(1, len(bn_names) // 40)]
    blob["bn_names"] = np.array(pick)
    for i, k in enumerate(pick):
        blob["bn%d" % i] = sd[k].numpy()
    blob["bn_tracked"] = np.asarray(int(sd["backbone.0.rbr_dense.bn.num_batches_tracked"]))
    blob["size"] = np.asarray(SIZE)
    blob["targets"] = targets.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_%s%s.npz" % (scale, suffix)), **blob)
    print("wrote train_%s%s.npz" % (scale, suffix), len(blob), "arrays")


if __name__ == "__main__":
    main()
