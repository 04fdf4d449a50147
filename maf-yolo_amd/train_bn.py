"""Training-mode BatchNorm2d (+ the activation behind it) on the HIP kernels: one BatchNorm (Conv.forward = act(bn(conv(x))), yolov6/layers/common.py:44-47) and the SUM of
the branch BatchNorms of a DilatedReparamBlock / RepVGGBlock (common.py:224, 3024-3031) as one apply pass per direction.

One of the four family files train_ops.py was cut into in round 6 (train_conv / train_dw / train_bn / train_cat).  `T` is train_ops itself: every module-level switch, cache and
helper lives THERE (tests, tools and tape.py read and set them as `train_ops.<name>`), and every reference from here goes through `T.<name>` at call time — so a switch flipped or
an entry point replaced on train_ops (bench.py --torch-convs) reaches this code exactly as it did when all of it was one file.  train_ops re-exports everything defined here;
import train_ops (or the package), not this file."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import lib, pack
from . import train_ops as T


def _bn_part(dev, c):
    """(scratch, phase): [2][R][2][roundup(c,256)] fp32, zeroed when it is allocated; a BatchNorm call accumulates into half `phase` and clears
    the other one (csrc/bn_act.hip), so the phase alternates per call on a buffer — kernels on one stream are ordered, different streams get
    different buffers."""
    if T._rec is not None:
        return T._tzeros(2 * T._BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    key = (dev.index, T._stream(dev), -(-c // 256))
    ent = T._bn_scratch.get(key)
    if ent is None:
        if len(T._bn_scratch) > 64:
            T._bn_scratch.clear()
        ent = T._bn_scratch[key] = [T._tzeros(2 * T._BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1]
    ent[1] ^= 1
    return ent[0], ent[1]


@T._laned
class _BNAct(torch.autograd.Function):
    """act(BatchNorm2d(x) [+ residual]) in training mode on the HIP kernels of csrc/bn_act.hip (batch statistics, running-stat update)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act, counter=None, residual=None, pre_stats=None, out=None):
        x, xs = T.nhwc(x)
        B, c, H, W = x.shape
        dt = T._DT[x.dtype]
        dev = x.device
        # out: the caller's slot of a concat buffer (an NHWC channel slice, cat_buffer below): the apply pass stores there and the cat never runs
        y = T._empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last) if out is None else out[0]      # (a tuple: not an input of the autograd node)
        stat = T._empty(2, c, dtype=torch.float32, device=dev)                 # save_mean, save_rstd
        sp = stat.data_ptr()                                                      # (pointer arithmetic: indexing a tensor costs ~2 us of host time, 4 per call)
        part, phase = T._bn_part(dev, c) if pre_stats is None else pre_stats       # pre_stats: (scratch, phase) whose half the producer of x has filled
        g32 = gamma.detach() if gamma.dtype == torch.float32 and gamma.is_contiguous() else gamma.detach().float().contiguous()
        b32 = beta.detach() if beta.dtype == torch.float32 and beta.is_contiguous() else beta.detach().float().contiguous()
        rs = 0
        if residual is not None:
            residual, rs = T.nhwc(residual)
        npass = 3 if residual is None else 4
        with T._prof("bn_act_forward", npass * B * H * W * c * x.element_size(), dev, (B, H, W, c, xs, act)):    # statistics pass (read) + apply pass (read [, read], write)
            lib.check(lib.load().maf_bn_forward_ex(x.data_ptr(), xs, B * H * W, c, dt, g32.data_ptr(), b32.data_ptr(), float(eps), float(momentum),
                                                None if running_mean is None else running_mean.data_ptr(),
                                                None if running_var is None else running_var.data_ptr(),
                                                None if counter is None else counter.data_ptr(), act,
                                                y.data_ptr(), y.stride()[3], sp, sp + 4 * c, part.data_ptr(), T._BN_REPLICAS,
                                                   phase, None if residual is None else residual.data_ptr(), rs, 0 if pre_stats is None else 1, T._stream(dev)))
        ctx.has_res = residual is not None
        ctx.res_in_bwd = residual is not None and act != lib.ACT_NONE          # the activation's derivative needs u = BN(x) + residual
        if ctx.res_in_bwd:
            ctx.save_for_backward(x, g32, b32, stat, residual)
        else:
            ctx.save_for_backward(x, g32, b32, stat)
        ctx.act = act
        # the Parameters themselves (not saved tensors: they are inputs of this node): backward adds dgamma / dbeta straight into their slices of a gradient exchange
        ctx.affine = (gamma, beta) if isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) else None
        T.stats["native_bn_act"] = T.stats.get("native_bn_act", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dz):
        if ctx.res_in_bwd:
            x, g32, b32, stat, residual = ctx.saved_tensors
        else:
            (x, g32, b32, stat), residual = ctx.saved_tensors, None
        B, c, H, W = x.shape
        dz, dzs = T.nhwc(dz)
        if dz.dtype != x.dtype:
            dz = dz.to(x.dtype)
            dzs = dz.stride()[3]
        x, xs = T.nhwc(x)
        dev = x.device
        dx = T._empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        # with a gradient exchange the apply kernel ADDS dgamma / dbeta to the parameters' bucket slices (main stream) and autograd gets None:
        # no AccumulateGrad add kernel per affine parameter (280 launches per step of n)
        from . import exchange
        ex, tg, tb = exchange.current, None, None
        if ex is not None and ctx.affine is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and T.bn_affine_direct:
            tg, tb = ex.target(ctx.affine[0]), ex.target(ctx.affine[1])
        direct = tg is not None and tb is not None and tg[1].is_contiguous() and tb[1].is_contiguous()
        dgb = None if direct else T._empty(2, c, dtype=torch.float32, device=dev)                  # dgamma, dbeta
        part, phase = T._bn_part(dev, c)
        dres, rs = None, 0
        if residual is not None:
            residual, rs = T.nhwc(residual)
            dres = T._empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        npass = 5 if residual is None else 8
        with T._prof("bn_act_backward", npass * B * H * W * c * x.element_size(), dev, (B, H, W, c, xs, dzs, ctx.act)):      # reduction pass (x, dz read) + apply pass (x, dz read, dx written)
            lib.check(lib.load().maf_bn_backward_acc(x.data_ptr(), xs, dz.data_ptr(), dzs, B * H * W, c, T._DT[x.dtype], g32.data_ptr(), b32.data_ptr(),
                                                     stat.data_ptr(), stat.data_ptr() + 4 * c, ctx.act, dx.data_ptr(), dx.stride()[3],
                                                     tg[1].data_ptr() if direct else dgb.data_ptr(), tb[1].data_ptr() if direct else dgb.data_ptr() + 4 * c,
                                                     part.data_ptr(), T._BN_REPLICAS, phase,
                                                     None if residual is None else residual.data_ptr(), rs,
                                                     None if dres is None else dres.data_ptr(), 0 if dres is None else dres.stride()[3], 1 if direct else 0, T._stream(dev)))
        if ctx.has_res and dres is None:
            dres = dz                                                            # no activation: the residual's gradient is dz itself
        if direct:
            ex.main_done(ctx.affine[0])
            ex.main_done(ctx.affine[1])
            return dx, None, None, None, None, None, None, None, None, dres, None, None
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, dres, None, None


def bn_act(x, bn, act=None, residual=None, pre_stats=None, out=None):
    """act(bn(x) [+ residual]) for an nn.BatchNorm2d `bn` and act in {None, 'relu', 'silu'}.  `out`: a slot of a concat buffer (`CatBuffer.slot`; HIP path only)
    the result is stored into — and returned as.  `pre_stats`: what dw_branches returned for this branch (its
    kernel has accumulated the batch statistics already: apply pass only), else None.  Training mode on CUDA tensors runs the fused HIP
    kernels (one statistics pass + one normalise/affine/[add]/activation pass; backward likewise); eval mode and CPU tensors run torch ops.
    `residual` (same shape as x; act None or 'relu'): the branch sums of RepVGGBlock / DilatedReparamBlock without a pass of their own."""
    mult = 8 if x.dtype == torch.float16 else 4
    if not (x.is_cuda and bn.training) or T.framework_ops:
        # CPU tensors (CI / gloo tests) and eval-mode BatchNorm inside a train-form forward (Model.forward(val_loss=True) never comes here:
        # it runs the deploy engine): torch ops, counted so that an A/B on `stats` cannot mistake them for the HIP path
        T.stats["torch_bn"] = T.stats.get("torch_bn", 0) + 1
        if out is not None:
            raise lib.MafError("bn_act: out= is a feature of the HIP path (the caller checks `cat_free_ok`)")
        y = bn(x)
        if residual is not None:
            y = y + residual
        return y if act in (None, "none") else (F.relu(y) if act == "relu" else F.silu(y))
    if not (x.dtype in T._DT and x.dim() == 4 and x.shape[1] % mult == 0 and bn.affine):
        raise lib.MafError("bn_act: unsupported input for the HIP path: %s %s (channels must be a multiple of %d, affine BatchNorm)" % (tuple(x.shape), x.dtype, mult))
    if residual is not None:
        if act == "silu":
            raise lib.MafError("bn_act: a residual goes with act None or 'relu'")
        if residual.shape != x.shape:
            raise lib.MafError("bn_act: residual %s must have the shape of x %s" % (tuple(residual.shape), tuple(x.shape)))
        if residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
    counter = bn.num_batches_tracked if bn.track_running_stats else None        # += 1 inside the apply kernel (140 one-element launches per step otherwise)
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64):
        counter.add_(1)
        counter = None
    if bn.momentum is None and bn.track_running_stats:
        # nn.BatchNorm2d(momentum=None) = cumulative moving average (factor 1 / num_batches_tracked): not what the kernel implements, and
        # not what the reference builds (momentum 0.03, yolov6/utils/torch_utils.py:43-45) — refuse rather than freeze the statistics
        raise lib.MafError("bn_act: BatchNorm2d(momentum=None) (cumulative average) is not supported on the HIP path")
    momentum = 0.0 if bn.momentum is None else bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if out is not None and not (out.shape == x.shape and out.dtype == x.dtype and out.device == x.device and T.nhwc(out)[0] is out):
        raise lib.MafError("bn_act: out= must be an NHWC (channel-slice) view of x's shape and dtype")
    return T._BNAct.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, momentum, T._ACT[act], counter, residual, pre_stats, None if out is None else (out,))


def _phase_array(phases):
    arr = T._INT4(*[int(p_) for p_ in phases])
    if T._rec is not None:
        T._rec.toggle_array(arr, [(j, 1, 4) for j, p_ in enumerate(phases) if isinstance(p_, lib.Phase)])
    return arr


def _bnsum_part(dev, c, nb):
    """(scratch, phase) of maf_bn_sum_backward: [2][R][1 + nb][roundup(c,256)] fp32 per (stream, width, branch count), halves alternating call by call."""
    if T._rec is not None:
        return T._tzeros(2 * T._BN_REPLICAS * (1 + nb) * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    key = (dev.index, T._stream(dev), -(-c // 256), nb)
    ent = T._bnsum_scratch.get(key)
    if ent is None:
        if len(T._bnsum_scratch) > 64:
            T._bnsum_scratch.clear()
        ent = T._bnsum_scratch[key] = [T._tzeros(2 * T._BN_REPLICAS * (1 + nb) * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1]
    ent[1] ^= 1
    return ent[0], ent[1]


@T._laned
class _BNSum(torch.autograd.Function):
    """sum_j BatchNorm2d_j(z_j) in training mode, no activation (the branch sum of a DilatedReparamBlock, yolov6/layers/common.py:3024-3031) on csrc/bn_sum.hip:
    ONE apply pass forward (the statistics come from the depth-wise kernel's epilogue or a statistics launch per branch that lacks them), one statistics + one
    apply launch backward for ALL branches (their upstream gradient is the same tensor)."""

    @staticmethod
    def forward(ctx, nb, cfg, *t):
        """t = z_0 .. z_{nb-1}, gamma_0 .. gamma_{nb-1}, beta_0 .. beta_{nb-1}; cfg = (eps, momentum, [(running_mean, running_var, counter, scratch, phase, need_stats)] per branch)"""
        zs = [T.nhwc(z) for z in t[:nb]]
        gammas, betas = t[nb:2 * nb], t[2 * nb:3 * nb]
        x0 = zs[0][0]
        B, c, H, W = x0.shape
        dt = T._DT[x0.dtype]
        dev = x0.device
        eps, momentum, per, act, dst, nxt = cfg                                 # nxt: (scratch, phase) of the BatchNorm that normalises the sum next — this pass accumulates its statistics — or None
        L = lib.load()
        M_ = B * H * W
        for (z, zst), (rm, rv, cnt, part, phase, need) in zip(zs, per):
            if need:                                                             # this branch's producer has no statistics epilogue (the 1 x 1 scale branch)
                lib.check(L.maf_bn_stats(z.data_ptr(), zst, M_, c, dt, part.data_ptr(), T._BN_REPLICAS, phase, T._stream(dev)))
        out = T._empty((B, c, H, W), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) if dst is None else dst      # dst: a concat buffer's slot (bn_act's out=)
        stat = T._empty(nb, 2, c, dtype=torch.float32, device=dev)            # save_mean, save_rstd per branch
        sp = stat.data_ptr()
        g32 = [g.detach() if g.dtype == torch.float32 and g.is_contiguous() else g.detach().float().contiguous() for g in gammas]
        b32 = [b.detach() if b.dtype == torch.float32 and b.is_contiguous() else b.detach().float().contiguous() for b in betas]
        # (the argument arrays are built OUTSIDE the timed region: on a host-bound eager step the event pair would measure their construction)
        args = (T._PTR4(*[z.data_ptr() for z, _ in zs]), T._INT4(*[zst for _, zst in zs]), nb, M_, c, dt,
                T._PTR4(*[g.data_ptr() for g in g32]), T._PTR4(*[b.data_ptr() for b in b32]), float(eps), float(momentum),
                T._PTR4(*[0 if p[0] is None else p[0].data_ptr() for p in per]), T._PTR4(*[0 if p[1] is None else p[1].data_ptr() for p in per]),
                T._PTR4(*[0 if p[2] is None else p[2].data_ptr() for p in per]),
                out.data_ptr(), out.stride()[3], T._PTR4(*[sp + 8 * c * j for j in range(nb)]), T._PTR4(*[sp + 8 * c * j + 4 * c for j in range(nb)]),
                T._PTR4(*[p[3].data_ptr() for p in per]), T._BN_REPLICAS, T._phase_array([p[4] for p in per]), act, T._stream(dev))
        with T._prof("bn_sum_forward", (nb + 1) * M_ * c * x0.element_size(), dev, (B, H, W, c, nb)):
            if nxt is None:
                lib.check(L.maf_bn_sum_forward(*args))
            else:
                lib.check(L.maf_bn_sum_forward_stats(*args[:-1], nxt[0].data_ptr(), T._BN_REPLICAS, nxt[1], args[-1]))
        ctx.save_for_backward(stat, *[z for z, _ in zs], *g32, *b32)
        ctx.nb, ctx.act = nb, act
        ctx.affine = list(zip(gammas, betas)) if all(isinstance(g, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter) for g, b in zip(gammas, betas)) else None
        T.stats["native_bn_act"] = T.stats.get("native_bn_act", 0) + nb
        T.stats["native_bn_sum"] = T.stats.get("native_bn_sum", 0) + 1
        return out

    @staticmethod
    def backward(ctx, dy):
        nb = ctx.nb
        sv = ctx.saved_tensors
        stat, zs, g32, b32 = sv[0], [T.nhwc(z) for z in sv[1:1 + nb]], sv[1 + nb:1 + 2 * nb], sv[1 + 2 * nb:1 + 3 * nb]
        x0 = zs[0][0]
        B, c, H, W = x0.shape
        dev = x0.device
        dy, dys = T.nhwc(dy)
        if dy.dtype != x0.dtype:
            dy = dy.to(x0.dtype)
            dys = dy.stride()[3]
        dzs = [T._empty((B, c, H, W), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) for _ in range(nb)]
        from . import exchange
        ex, tg = exchange.current, None
        if ex is not None and ctx.affine is not None and T.bn_affine_direct and all(ctx.needs_input_grad[2 + nb + j] and ctx.needs_input_grad[2 + 2 * nb + j] for j in range(nb)):
            tg = [(ex.target(g), ex.target(b)) for g, b in ctx.affine]
            if not all(a_ is not None and b_ is not None and a_[1].is_contiguous() and b_[1].is_contiguous() for a_, b_ in tg):
                tg = None
        dgb = None if tg is not None else T._empty(nb, 2, c, dtype=torch.float32, device=dev)
        part, phase = T._bnsum_part(dev, c, nb)
        sp = stat.data_ptr()
        gp = None if dgb is None else dgb.data_ptr()
        args = (dy.data_ptr(), dys, T._PTR4(*[z.data_ptr() for z, _ in zs]), T._INT4(*[zst for _, zst in zs]), nb, B * H * W, c, T._DT[x0.dtype],
                                                     T._PTR4(*[g.data_ptr() for g in g32]), T._PTR4(*[b.data_ptr() for b in b32]), T._PTR4(*[sp + 8 * c * j for j in range(nb)]), T._PTR4(*[sp + 8 * c * j + 4 * c for j in range(nb)]),
                                                     T._PTR4(*[d.data_ptr() for d in dzs]), T._INT4(*[d.stride()[3] for d in dzs]),
                                                     T._PTR4(*[tg[j][0][1].data_ptr() if tg is not None else gp + 8 * c * j for j in range(nb)]),
                                                     T._PTR4(*[tg[j][1][1].data_ptr() if tg is not None else gp + 8 * c * j + 4 * c for j in range(nb)]),
                                                     1 if tg is not None else 0, part.data_ptr(), T._BN_REPLICAS, phase, ctx.act, T._stream(dev))
        with T._prof("bn_sum_backward", (2 + 3 * nb) * B * H * W * c * x0.element_size(), dev, (B, H, W, c, nb)):
            lib.check(lib.load().maf_bn_sum_backward(*args))
        if tg is not None:
            for g, b in ctx.affine:
                ex.main_done(g)
                ex.main_done(b)
            return (None, None, *dzs, *([None] * (2 * nb)))
        return (None, None, *dzs, *[dgb[j, 0] for j in range(nb)], *[dgb[j, 1] for j in range(nb)])


def bn_sum(zs, bns, pre_stats=None, act=None, out=None, next_bn=None):
    """act(sum_j bns[j](zs[j])): the branches of a train-form DilatedReparamBlock (act None) or of a RepVGGBlock (act "relu", common.py:224).  CUDA + training mode: csrc/bn_sum.hip (one apply pass forward,
    statistics + apply for all branches backward); otherwise — and for anything the kernel does not take — the chain of bn_act calls with `residual`.
    `pre_stats[j]`: what dw_branches returned for branch j (its statistics are already accumulated) or None.
    `next_bn`: the BatchNorm2d that normalises the result next (UniRepLKNetBlock.norm): returns (result, pre_stats for bn_act(result, next_bn, ...)) — the apply pass has
    accumulated that BatchNorm's batch statistics (csrc/bn_sum.hip, STATS form) — or (result, None) where it cannot."""
    nb = len(zs)
    pre = list(pre_stats) if pre_stats is not None else [None] * nb
    x = zs[0]
    mult = 8 if x.dtype == torch.float16 else 4
    ok = (T.bn_sum_merged and 2 <= nb <= 4 and x.is_cuda and not T.framework_ops and not T._deterministic and x.dtype in T._DT and x.dim() == 4 and x.shape[1] % mult == 0
          and all(z.shape == x.shape and z.dtype == x.dtype for z in zs)
          and all(bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.eps == bns[0].eps and bn.momentum == bns[0].momentum
                  and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64 for bn in bns))
    if act not in (None, "none", "relu"):
        raise lib.MafError("bn_sum: act must be None or 'relu'")
    if not ok:
        y = T.bn_act(zs[0], bns[0], pre_stats=pre[0])
        for j in range(1, nb):
            y = T.bn_act(zs[j], bns[j], act if j == nb - 1 else None, residual=y, pre_stats=pre[j], out=out if j == nb - 1 else None)
        return y if next_bn is None else (y, None)
    if out is not None and not (out.shape == x.shape and out.dtype == x.dtype and out.device == x.device and T.nhwc(out)[0] is out):
        raise lib.MafError("bn_sum: out= must be an NHWC (channel-slice) view of the branches' shape and dtype")
    per = []
    for bn, st in zip(bns, pre):
        part, phase = st if st is not None else T.bn_own_scratch(bn, x.device, x.shape[1])
        per.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, part, phase, st is None))
    nxt = None
    if (next_bn is not None and T.bn_sum_next_stats and next_bn.training and next_bn.affine and next_bn.track_running_stats and next_bn.momentum is not None
            and x.shape[1] // mult <= 256):
        nxt = T.bn_own_scratch(next_bn, x.device, x.shape[1])
        T.stats["bn_sum_next_stats"] = T.stats.get("bn_sum_next_stats", 0) + 1
    y = T._BNSum.apply(nb, (bns[0].eps, bns[0].momentum, per, T._ACT[act], out, nxt), *zs, *[bn.weight for bn in bns], *[bn.bias for bn in bns])
    return y if next_bn is None else (y, nxt)
