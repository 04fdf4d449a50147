"""The optimizer of the train step (BASELINE.json north_star: "... + SGD step"), with the parameter grouping of the reference's
build_optimizer (yolov6/solver/build.py:12-33): BatchNorm weights without weight decay, every other weight with it, biases without.

    opt = build_optimizer(model, lr0=0.01, momentum=0.937, weight_decay=5e-4)          # configs/MAF-YOLO-n.py:19-29

On CUDA parameters the SGD is torch's fused implementation: one multi-tensor launch per group, and — what matters for the step time — the
GradScaler's inf check stays on the device (the optimizer receives grad_scale / found_inf tensors and skips the update itself), so
`scaler.step(opt)` does not synchronise the host with the GPU and the launches of step i+1 queue up behind step i.  Same arithmetic as the
reference's torch.optim.SGD(nesterov=True)."""
import torch
import torch.nn as nn


def param_groups(model):
    """(BatchNorm weights, other weights, biases) in module order — build.py:14-21."""
    bn_w, w, b = [], [], []
    for m in model.modules():
        bias = getattr(m, "bias", None)
        if isinstance(bias, nn.Parameter):
            b.append(bias)
        weight = getattr(m, "weight", None)
        if isinstance(m, nn.BatchNorm2d):
            bn_w.append(m.weight)
        elif isinstance(weight, nn.Parameter):
            w.append(weight)
    return bn_w, w, b


def build_optimizer(model, lr0=0.01, momentum=0.937, weight_decay=5e-4, optim="SGD", fused=None):
    """torch.optim.SGD(nesterov=True) / Adam over the three groups of `param_groups` (build.py:23-30).  fused=None: the fused
    implementation when every parameter lives on a CUDA device."""
    bn_w, w, b = param_groups(model)
    if fused is None:
        fused = all(p.is_cuda for p in bn_w + w + b) and len(bn_w + w + b) > 0
    if optim == "SGD":
        opt = torch.optim.SGD(bn_w, lr=lr0, momentum=momentum, nesterov=True, fused=fused)
    elif optim == "Adam":
        opt = torch.optim.Adam(bn_w, lr=lr0, betas=(momentum, 0.999), fused=fused)
    else:
        raise ValueError("unknown optimizer %r (SGD / Adam)" % (optim,))
    opt.add_param_group({"params": w, "weight_decay": weight_decay})
    opt.add_param_group({"params": b})
    return opt
