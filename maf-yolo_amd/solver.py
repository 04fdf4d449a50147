"""The optimizer and the weight average of the train step (BASELINE.json north_star: "... + SGD step"), with the parameter grouping of the reference's
build_optimizer (yolov6/solver/build.py:12-33): BatchNorm weights without weight decay, every other weight with it, biases without.

    opt = build_optimizer(model, lr0=0.01, momentum=0.937, weight_decay=5e-4)          # configs/MAF-YOLO-n.py:19-29

On CUDA parameters the optimizer is NativeSGD — a subclass of torch's fused SGD whose step is ONE launch over a device table of every parameter
(csrc/train_ops.hip:sgd_update_kernel), bit-identical to the fused implementation — and, what matters most for the step time, the GradScaler's
inf check stays on the device (the optimizer receives grad_scale / found_inf tensors and skips the update itself), so `scaler.step(opt)` does
not synchronise the host with the GPU and the launches of step i+1 queue up behind step i.  `GradScaler` here is torch.amp.GradScaler with
that inf check as one launch over the contiguous gradient ranges.  Same arithmetic as the reference's torch.optim.SGD(nesterov=True) up to the
fused kernel's double-precision hyper-parameter arithmetic (the framework's own fused = True path)."""
import os

from .config import cfg

import torch
import torch.nn as nn

ema_one_launch = cfg.ema_native       # A/B switch: ModelEMA.update as one launch (csrc/train_ops.hip maf_ema_update)
sgd_one_launch = cfg.sgd_native       # A/B switch: the SGD step as one launch (csrc/train_ops.hip maf_sgd_update)


inf_check_one_launch = cfg.inf_check_native       # A/B switch: GradScaler's inf check as one launch (maf_nonfinite_check)


class GradScaler(torch.amp.GradScaler):
    """torch.amp.GradScaler (the reference's `amp.GradScaler`, yolov6/core/engine.py:297, :375-391) whose inf check in front of an optimizer step that takes the scale itself
    (`_step_supports_amp_scaling`: the fused / native SGD) is ONE launch over the contiguous ranges the gradients occupy (csrc/train_ops.hip:nonfinite_check_kernel,
    maf_nonfinite_check) — the flat buckets of a GradExchange are a handful of ranges for ~300 tensors — instead of the framework's multi-tensor launches (four per step,
    64 us, on MAF-YOLO-n).  Same `found_inf` (0 / 1) on the device, same scale update; gradients that are not dense fp32 CUDA tensors on the scale's device, or an explicit
    `unscale_()`, take the parent's path."""

    def __init__(self, device="cuda", **kwargs):
        super().__init__(device, **kwargs)
        self._maf_ranges = {}

    def _maf_table(self, optimizer, dev):
        grads = []
        for group in optimizer.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (g.is_cuda and g.device == dev and g.dtype == torch.float32 and not g.is_sparse and g.is_contiguous()):
                    return None
                if g.numel():
                    grads.append((g.data_ptr(), g.numel()))
        if not grads:
            return None
        sig = tuple(grads)
        ent = self._maf_ranges.get(id(optimizer))
        if ent is None or ent[0] != sig:
            import ctypes as C
            from . import lib
            assert C.sizeof(lib.MafRangeDesc) == lib.load().maf_range_desc_size()
            merged = []
            for ptr, n in sorted(set(grads)):
                if merged and merged[-1][0] + 4 * merged[-1][1] == ptr:
                    merged[-1][1] += n
                elif merged and ptr < merged[-1][0] + 4 * merged[-1][1]:          # overlapping views: leave those to the framework
                    return None
                else:
                    merged.append([ptr, n])
            arr = (lib.MafRangeDesc * len(merged))()
            blk = 0
            for e, (ptr, n) in zip(arr, merged):
                e.ptr, e.total, e.block0 = ptr, n, blk
                blk += -(-n // 4096)
            ent = self._maf_ranges[id(optimizer)] = (sig, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev), len(merged), blk)
        return ent[1:]

    def _check_inf_per_device(self, optimizer):
        # (two private members of torch.amp.GradScaler are used below; a torch build that lacks either gets the parent's check — slower, same result)
        if not inf_check_one_launch or not hasattr(self, "_check_scale_growth_tracker") or not hasattr(self, "_per_optimizer_states"):
            return super()._check_inf_per_device(optimizer)
        _scale, _ = self._check_scale_growth_tracker("_check_inf_per_device")
        dev = _scale.device
        tab = self._maf_table(optimizer, dev) if dev.type == "cuda" else None
        if tab is None:
            return super()._check_inf_per_device(optimizer)
        from . import lib
        table, n, nblocks = tab
        found_inf = torch.full((), 0.0, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            lib.check(lib.load().maf_nonfinite_check(table.data_ptr(), n, nblocks, found_inf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        self._per_optimizer_states[id(optimizer)]["found_inf_per_device"] = {dev: found_inf}
        return self._per_optimizer_states[id(optimizer)]["found_inf_per_device"]


class NativeSGD(torch.optim.SGD):
    """torch.optim.SGD (constructed with fused=True: same state_dict, param_groups, schedulers, GradScaler protocol) whose step is ONE launch over a device
    descriptor table of every parameter with a gradient (csrc/train_ops.hip:sgd_update_kernel, include/mafyolo_hip.h:maf_sgd_update) instead of the framework's
    multi-tensor launches (10 per step on MAF-YOLO-n's 3 groups / ~300 tensors, 171 us at the serial tail of the step).  Bit-identical parameters and momentum
    buffers (the kernel follows the fused implementation operation by operation).  The table is rebuilt when an address in it changes (one pass over the
    parameter list per step, no device work); anything the kernel does not take — CPU / non-fp32 / sparse / non-contiguous tensors, dampening, maximize, a
    tensor learning rate, more than 8 groups — goes to the parent's step."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._maf_table = None
        self._maf_sig = None
        self.native_steps = 0

    def _maf_entries(self):
        ents = []
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0) != 0 or group.get("maximize", False) or isinstance(group["lr"], torch.Tensor) or group.get("differentiable", False):
                return None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                buf = self.state[p].get("momentum_buffer") if group["momentum"] != 0 else None
                if group["momentum"] != 0 and buf is None:                        # (torch's first step: buf = grad — build_optimizer pre-allocates zero buffers)
                    buf = self.state[p]["momentum_buffer"] = torch.zeros_like(p)
                ts = (p, g) + ((buf,) if buf is not None else ())
                if not all(t.is_cuda and t.dtype == torch.float32 and not t.is_sparse and t.is_contiguous() and t.device == p.device and t.numel() == p.numel() for t in ts):
                    return None
                if p.numel():
                    ents.append((p, g, buf, gi))
        return ents

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not sgd_one_launch or len(self.param_groups) > 8:
            return super().step(closure)
        ents = self._maf_entries()
        if not ents or len({e[0].device for e in ents}) != 1:
            return super().step(closure)
        import ctypes as C
        from . import lib
        dev = ents[0][0].device
        sig = tuple((p.data_ptr(), g.data_ptr(), b.data_ptr() if b is not None else 0, gi) for p, g, b, gi in ents)
        if sig != self._maf_sig:
            assert C.sizeof(lib.MafSgdDesc) == lib.load().maf_sgd_desc_size()
            arr = (lib.MafSgdDesc * len(ents))()
            blk = 0
            for e, (p, g, b, gi) in zip(arr, ents):
                e.param, e.grad, e.buf, e.total, e.block0, e.group = p.data_ptr(), g.data_ptr(), (b.data_ptr() if b is not None else None), p.numel(), blk, gi
                blk += -(-p.numel() // 1024)
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._maf_table, self._maf_sig = (table, len(ents), blk), sig
        table, n, nblocks = self._maf_table
        ng = len(self.param_groups)
        lr = (C.c_double * ng)(*[float(g_["lr"]) for g_ in self.param_groups])
        wd = (C.c_double * ng)(*[float(g_["weight_decay"]) for g_ in self.param_groups])
        mu = (C.c_double * ng)(*[float(g_["momentum"]) for g_ in self.param_groups])
        nest = (C.c_int32 * ng)(*[1 if g_["nesterov"] else 0 for g_ in self.param_groups])
        found_inf, grad_scale = getattr(self, "found_inf", None), getattr(self, "grad_scale", None)     # set by GradScaler.step (_step_supports_amp_scaling)
        for t in (found_inf, grad_scale):
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 and t.device == dev):
                return super().step(closure)
        with torch.cuda.device(dev):
            lib.check(lib.load().maf_sgd_update(table.data_ptr(), n, nblocks, ng, lr, wd, mu, nest,
                                                found_inf.data_ptr() if found_inf is not None else None, grad_scale.data_ptr() if grad_scale is not None else None,
                                                torch.cuda.current_stream(dev).cuda_stream))
        self.native_steps += 1
        return None


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
        opt = (NativeSGD if fused else torch.optim.SGD)(bn_w, lr=lr0, momentum=momentum, nesterov=True, fused=fused)
    elif optim == "Adam":
        opt = torch.optim.Adam(bn_w, lr=lr0, betas=(momentum, 0.999), fused=fused)
    else:
        raise ValueError("unknown optimizer %r (SGD / Adam)" % (optim,))
    opt.add_param_group({"params": w, "weight_decay": weight_decay})
    opt.add_param_group({"params": b})
    if fused and optim == "SGD" and momentum != 0:
        # torch's fused SGD allocates the momentum buffers with empty_like on its first call and lets the kernel fill them (buf = grad);
        # when the GradScaler skips that first step (found_inf — the normal start of an AMP run: the scale starts at 65536) the kernel
        # returns early and every later step reads uninitialised memory as momentum (MAF-YOLO-m went to NaN at its first unskipped step).
        # Zero buffers give the same arithmetic — first real step: buf = momentum * 0 + grad — without that hole.
        for group in opt.param_groups:
            for p in group["params"]:
                opt.state[p]["momentum_buffer"] = torch.zeros_like(p)
    return opt


class ModelEMA:
    """Exponential moving average of every floating-point entry of the model's state_dict (parameters AND buffers) — the reference's
    ModelEMA (yolov6/utils/ema.py:11-42), updated after every optimizer step on the main process (engine.py:67, :389-390) and the model
    that is evaluated / checkpointed (`self.ema.ema`, engine.py:198, :246).

    Same attributes (`ema`, `updates`, `decay`) and arithmetic — decay_t = decay * (1 - exp(-t / 2000)); e = e * decay_t + (1 - decay_t) * m,
    rounded in that order — but the update is three multi-tensor launches over all ~840 tensors instead of two small kernels per tensor
    (the reference's Python loop issues ~1 700 launches per step)."""

    def __init__(self, model, decay=0.9999, updates=0):
        import copy
        import math
        self.ema = copy.deepcopy(de_parallel(model)).eval()                       # fp32 copy, eval mode, no gradients
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._pairs_of = None
        self._sig = None

    def _pairs(self, model):
        src = de_parallel(model)
        stale = self._pairs_of is not src
        if not stale:
            # the cached tensors must still BE the model's / the average's storage: `.to()`, `.cuda()`, `p.data = ...` replace it silently.
            # One pass over the parameter lists comparing addresses (no device work) — the reference re-reads state_dict() on every update.
            stale = self._sig != self._signature(src)
        if stale:                                                                 # (ema tensor, model tensor) of every floating entry, state_dict order
            sd = src.state_dict()
            self._dst = [v for k, v in self.ema.state_dict().items() if v.dtype.is_floating_point]
            self._src = [sd[k].detach() for k, v in self.ema.state_dict().items() if v.dtype.is_floating_point]
            self._pairs_of = src
            self._plist = None                                                    # (the cached Parameter objects of _signature: rebuilt with the lists)
            self._table = None                                                    # (the device descriptor table of the one-launch update)
            self._sig = self._signature(src)
        return self._dst, self._src

    def _signature(self, src):
        """What must not have changed for the cached tensor lists to be the models' storage.  A maf_yolo_amd.Model counts its `_apply` calls (every
        .to() / .cuda() / .half() / .float() goes through it and is what replaces buffers and parameter storage): generation numbers + the addresses of
        the cached Parameter objects (~0.15 ms).  Any other module: the addresses of every parameter and buffer of both trees — four walks of the
        module tree, 3-4 ms of host time per update on MAF-YOLO-n, which the step (issue-bound on the host) paid in full until round 4."""
        gen, gen_e = getattr(src, "_maf_apply_gen", None), getattr(self.ema, "_maf_apply_gen", None)
        if gen is not None and gen_e is not None and self._pairs_of is src and getattr(self, "_plist", None) is not None:
            return ("gen", gen, gen_e, tuple([p.data_ptr() for p in self._plist]))
        self._plist = list(src.parameters()) + list(self.ema.parameters())
        if gen is not None and gen_e is not None:
            return ("gen", gen, gen_e, tuple([p.data_ptr() for p in self._plist]))
        return (tuple(p.data_ptr() for p in src.parameters()), tuple(b.data_ptr() for b in src.buffers()),
                tuple(p.data_ptr() for p in self.ema.parameters()), tuple(b.data_ptr() for b in self.ema.buffers()))

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            dst, src = self._pairs(model)
            if self._native(dst, src):
                from . import lib
                table, n, nblocks, dev = self._table
                with torch.cuda.device(dev):
                    lib.check(lib.load().maf_ema_update(table.data_ptr(), n, nblocks, d, 1 - d, torch.cuda.current_stream(dev).cuda_stream))
                return
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, torch._foreach_mul(src, 1 - d))

    def _native(self, dst, src):
        """All pairs fp32, dense, on one CUDA device: the update is ONE launch over a descriptor table (csrc/train_ops.hip maf_ema_update; same
        roundings as the three multi-tensor ops below it, which cost ~1.8 ms of host time per step on MAF-YOLO-n: ~840 tensors, a temporary
        per tensor).  The table is rebuilt with the cached tensor lists (`_pairs`).  Anything else (CPU models, half-precision copies) takes the
        framework's multi-tensor ops."""
        if getattr(self, "_table", None) is None:
            self._table = False
            if ema_one_launch and dst and all(a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.device == b.device == dst[0].device
                           and a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() for a, b in zip(dst, src)):
                import ctypes as C
                from . import lib
                pairs = [(a, b) for a, b in zip(dst, src) if a.numel()]
                arr = (lib.MafEmaDesc * len(pairs))()
                blk = 0
                for e, (a, b) in zip(arr, pairs):
                    e.dst, e.src, e.total, e.block0 = a.data_ptr(), b.data_ptr(), a.numel(), blk
                    blk += -(-a.numel() // 1024)
                assert C.sizeof(lib.MafEmaDesc) == lib.load().maf_ema_desc_size()
                if pairs:
                    self._table = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dst[0].device), len(pairs), blk, dst[0].device)
        return bool(self._table)

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        """ema.py:39-40 / copy_attr :43-49: plain attributes of the model copied onto the EMA model."""
        for k, v in de_parallel(model).__dict__.items():
            if (len(include) and k not in include) or k.startswith("_") or k in exclude:
                continue
            setattr(self.ema, k, v)


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def de_parallel(model):
    return model.module if is_parallel(model) else model
