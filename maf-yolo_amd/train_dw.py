"""Training-form depth-wise convolutions on the HIP kernels: a single k x k conv (UniRepLKNetBlock in the heads) and the parallel branches of a train-form
DilatedReparamBlock in one launch per direction, with their weight gradients (yolov6/layers/common.py:2948-3051, 3053-3100).

One of the four family files train_ops.py was cut into in round 6 (train_conv / train_dw / train_bn / train_cat).  `T` is train_ops itself: every module-level switch, cache and
helper lives THERE (tests, tools and tape.py read and set them as `train_ops.<name>`), and every reference from here goes through `T.<name>` at call time — so a switch flipped or
an entry point replaced on train_ops (bench.py --torch-convs) reaches this code exactly as it did when all of it was one file.  train_ops re-exports everything defined here;
import train_ops (or the package), not this file."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import lib, pack
from . import train_ops as T


def _launch_dw(x, xs, wp, bias, B, H, W, c, k, out, dt):
    ys = out.stride()[3]
    key = (2, dt, B, H, W, c, k, xs, ys)
    op = T._op_cache.get(key)
    if op is None:
        op = T._op_cache[key] = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_DWCONV, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc = B, H, W, c, c, k, 1
        op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = c, xs, 0, lib.SRC_DIRECT
        op.out_stride, op.out_coff = ys, 0
    op.src[0].ptr, op.out, op.w, op.bias = x.data_ptr(), out.data_ptr(), wp.data_ptr(), bias.data_ptr()
    if T.profile is None:
        lib.check(lib.load().maf_op_launch(C.byref(op), T._stream(x.device)))
        return
    with T._prof("dwconv_k%d" % k, 2 * B * H * W * c * x.element_size(), x.device):
        lib.check(lib.load().maf_op_launch(C.byref(op), T._stream(x.device)))


def _packed_dw(w, c, k, flip, dt, dev):
    hit = T._hit(w, ("w", c, k, flip, dt))
    if hit is not None:
        return hit
    nbytes = c * k * k * (2 if dt == lib.F16 else 4)

    def now(dst):
        wf = w.detach().reshape(c, k * k).float().contiguous()
        lib.check(lib.load().maf_pack_dw(wf.data_ptr(), c, k, flip, dt, dst.data_ptr(), T._stream(dev)))

    if not (w.dtype == torch.float32 and w.is_contiguous() and w.is_leaf):
        buf = T._empty(nbytes, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    fields = dict(kind=1, dtype=dt, Cout=c, Cin=1, taps=k * k, transpose=0, CT=0, steps=0, Kp=0, flip=flip, total=c * k * k)
    return T._staged(w, ("w", c, k, flip, dt), nbytes, fields, now)


@T._laned
class _DWConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, xs = T.nhwc(x)
        B, c, H, W = x.shape
        k = w.shape[-1]
        dt = T._DT[x.dtype]
        out = T._empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        T._launch_dw(x, xs, T._packed_dw(w, c, k, 0, dt, x.device), T._zero_bias(x.device, c), B, H, W, c, k, out, dt)
        ctx.save_for_backward(x, w)
        T.stats["native_dwconv"] += 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, c, H, W = x.shape
        k = w.shape[-1]
        dy, dys = T.nhwc(dy)
        if dy.dtype != x.dtype:
            T._glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = T._DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            # one copy of dW: the kernel adds one value per (channel, tap) and workgroup after its own LDS reduction, so the replicas the
            # first version spread its atomics over (and the torch sum behind them) buy <= 7 % on the 160 x 160 layers and nothing elsewhere
            dw = T._dw_wgrad(x, dy, dys, w)
        if ctx.needs_input_grad[0]:                                              # correlation with the flipped kernel
            dx = T._empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            T._launch_dw(dy, dys, T._packed_dw(w, c, k, 1, dt, x.device), T._zero_bias(x.device, c), B, H, W, c, k, dx, dt)
        T._side_done(x.device, dw is not None)
        return dx, dw


def _dw_wgrad(x, dy, dys, w):
    """Weight gradient of a depth-wise conv on the side stream (csrc/train_ops.hip: dw_wgrad_kernel): None when it went into a gradient exchange's
    bucket slice, else dW like w."""
    B, c, H, W = x.shape
    k = w.shape[-1]
    dt = T._DT[x.dtype]
    xx, xs = T.nhwc(x)
    ex, view = T._grad_sink(w)
    L = lib.load()
    if ex is not None:                                                        # [C][k*k] is the parameter's layout: the atomics land in its bucket slice
        dwf = view
        h = T._fork(x.device, xx, dy)
    else:
        dwf = T._empty(c, k * k, dtype=torch.float32, device=x.device)
        h = T._fork(x.device, xx, dy, dwf)
        lib.check(L.maf_zero(dwf.data_ptr(), dwf.numel() * 4, h))
    with T._prof("dw_wgrad_k%d" % k, 2 * B * H * W * c * x.element_size(), x.device, (B, H, W, c, k, xs, dys), h):
        lib.check(L.maf_dw_wgrad(xx.data_ptr(), xs, dy.data_ptr(), dys, B, H, W, c, k, dt, dwf.data_ptr(), 1, h))
    if ex is not None:
        ex.side_done(w)
        return None
    return dwf.reshape(w.shape).to(w.dtype)


def _dw_wgrad31_ok(x, ws):
    """The branches the merged launch takes: kernel sizes (.., 3, 1) behind an optional larger first branch, on the maps where maf_dw_wgrad routes k = 3 to the
    vector kernel (csrc/train_ops.hip: maf_dw_wgrad; the small maps' k = 3 gradients run on the matrix cores)."""
    if not (T.dw_wgrad31 and x.is_cuda and x.dtype in T._DT):
        return None
    ks = tuple(int(w.shape[-1]) for w in ws)
    B, c, H, W = x.shape
    if W <= 96 and (H * W <= 400 or (H * W <= 1600 and c <= 192)):
        return None
    if ks == (3, 3, 1):
        return (0, 1, 2)
    if len(ks) == 3 and ks[1:] == (3, 1):
        return (1, None, 2)
    return None


def _dw_wgrad31(x, dzs, ws, sel):
    """maf_dw_wgrad31 on the side stream for the branches `sel` = (3x3, second 3x3 or None, 1x1): [dW or None (went into an exchange bucket)] per selected branch."""
    B, c, H, W = x.shape
    dt = T._DT[x.dtype]
    xx, xs = T.nhwc(x)
    L = lib.load()
    js = [j for j in sel if j is not None]
    sinks = {j: T._grad_sink(ws[j]) for j in js}
    bufs = {}
    for j in js:
        if sinks[j][0] is not None:
            bufs[j] = sinks[j][1]
        else:
            k = ws[j].shape[-1]
            bufs[j] = T._empty(c, k * k, dtype=torch.float32, device=x.device)
    own = [bufs[j] for j in js if sinks[j][0] is None]
    h = T._fork(x.device, xx, *[dzs[j] for j in js], *own)
    for t in own:
        lib.check(L.maf_zero(t.data_ptr(), t.numel() * 4, h))
    a, b, one = sel
    with T._prof("dw_wgrad_k31", (1 + len(js)) * B * H * W * c * x.element_size(), x.device, (B, H, W, c, len(js), xs), h):
        lib.check(L.maf_dw_wgrad31(xx.data_ptr(), xs, dzs[a].data_ptr(), dzs[a].stride()[3],
                                   None if b is None else dzs[b].data_ptr(), 0 if b is None else dzs[b].stride()[3],
                                   dzs[one].data_ptr(), dzs[one].stride()[3], B, H, W, c, dt,
                                   bufs[a].data_ptr(), None if b is None else bufs[b].data_ptr(), bufs[one].data_ptr(), 1, h))
    out = {}
    for j in js:
        if sinks[j][0] is not None:
            sinks[j][0].side_done(ws[j])
            out[j] = None
        else:
            out[j] = bufs[j].reshape(ws[j].shape).to(ws[j].dtype)
    T.stats["native_dw_wgrad31"] = T.stats.get("native_dw_wgrad31", 0) + 1
    return out


def _launch_dwb(srcs, dsts, wps, k0, B, H, W, c, dt, dgrad, dev, bstats=None):
    nb = len(wps)
    sp, ss = T._PTR4(*[t.data_ptr() for t in srcs]), T._INT4(*[t.stride()[3] for t in srcs])
    dp, ds = T._PTR4(*[t.data_ptr() for t in dsts]), T._INT4(*[t.stride()[3] for t in dsts])
    wp = T._PTR4(*[t.data_ptr() for t in wps])
    es = 2 if dt == lib.F16 else 4
    L = lib.load()
    with T._prof("dw_branches_dgrad_k%d" % k0 if dgrad else "dw_branches_k%d" % k0, (nb + 1) * B * H * W * c * es, dev, (B, H, W, c, k0, nb)):
        if bstats is not None:                                                   # [(scratch, phase)] per branch: the half its BatchNorm call will read
            half = T._BN_REPLICAS * 2 * (-(-c // 256) * 256)
            stp = T._PTR4(*[0 if st is None else st[0].data_ptr() + 4 * st[1] * half for st in bstats])
            if T._rec is not None:                                                 # the half alternates from replay to replay: the pointer words toggle between the two
                T._rec.toggle_array(stp, [(j, st[0].data_ptr() ^ (st[0].data_ptr() + 4 * half), 8) for j, st in enumerate(bstats) if st is not None])
            lib.check(L.maf_dw_branches_stats(sp, ss, dp, ds, wp, nb, k0, B, H, W, c, dt, stp, L.maf_bn_replicas(c, T._BN_REPLICAS), T._stream(dev)))
        else:
            lib.check(L.maf_dw_branches(sp, ss, dp, ds, wp, nb, k0, B, H, W, c, dt, 1 if dgrad else 0, T._stream(dev)))


@T._laned
class _DWBranches(torch.autograd.Function):
    """The parallel depth-wise branches of a train-form DilatedReparamBlock (yolov6/layers/common.py:3024-3031) on csrc/dw_branches.hip: one launch
    computes every branch's convolution of the shared input, one launch their summed data gradient; the weight gradients stay per branch on the side stream."""

    @staticmethod
    def forward(ctx, x, bstats, *ws):
        x, xs = T.nhwc(x)
        B, c, H, W = x.shape
        dt = T._DT[x.dtype]
        dev = x.device
        outs = [T._empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last) for _ in ws]
        T._launch_dwb([x], outs, [T._packed_dw(w, c, w.shape[-1], 0, dt, dev) for w in ws], ws[0].shape[-1], B, H, W, c, dt, False, dev, bstats)
        ctx.save_for_backward(x, *ws)
        T.stats["native_dwconv"] += len(ws)
        T.stats["native_dw_branches"] = T.stats.get("native_dw_branches", 0) + 1
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, *ws = ctx.saved_tensors
        B, c, H, W = x.shape
        dt = T._DT[x.dtype]
        dev = x.device
        dzs = []
        for dy in dys:
            if dy is None:                                                       # a branch nobody used (not in the reference's graph): zero gradient
                T._glue()
                dy = T._tzeros((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            dy, _ = T.nhwc(dy)
            dzs.append(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        dws = [None] * len(ws)
        returned = False
        merged = {}
        sel = T._dw_wgrad31_ok(x, ws) if all(ctx.needs_input_grad[2:2 + len(ws)]) else None
        if sel is not None:
            merged = T._dw_wgrad31(x, dzs, ws, sel)
        for j, w in enumerate(ws):
            if j in merged:
                dws[j] = merged[j]
            elif ctx.needs_input_grad[2 + j]:
                dws[j] = T._dw_wgrad(x, dzs[j], dzs[j].stride()[3], w)
            returned = returned or dws[j] is not None
        dx = None
        if ctx.needs_input_grad[0]:                                              # sum over the branches of the correlation with the flipped kernel
            dx = T._empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            T._launch_dwb(dzs, [dx], [T._packed_dw(w, c, w.shape[-1], 1, dt, dev) for w in ws], ws[0].shape[-1], B, H, W, c, dt, True, dev)
        T._side_done(dev, returned)
        return (dx, None, *dws)


def bn_own_scratch(bn, dev, c):
    """(scratch, phase) of a BatchNorm whose statistics are produced by ANOTHER kernel than its own call (the depth-wise kernel of csrc/dw_branches.hip):
    a buffer per module — the shared per-stream one alternates its halves call by call, and the apply pass of the call in front would clear the half this
    call's producer has just filled — whose halves alternate step by step (the apply pass clears the half of the step before, as always)."""
    if T._rec is not None:                                                         # a recording step tape: a scratch of its own per call site, the phase a toggled word
        return T._tzeros(2 * T._BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    ent = T._own_scratch.get(bn)
    if ent is None or ent[0].device != dev or ent[2] != c:
        ent = T._own_scratch[bn] = [T._tzeros(2 * T._BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1, c]
    ent[1] ^= 1
    return ent[0], ent[1]


def dw_branches(x, ws, bns=None):
    """[depth-wise conv of x with w for w in ws] for the k > 1 branches of a DilatedReparamBlock (kernel sizes k0, k0 - 2, ... 3; k0 = 3: 3, 3): ONE
    launch forward and one for the summed data gradient on CUDA tensors (csrc/dw_branches.hip); any other combination runs branch by branch.
    `bns` (the BatchNorm2d behind every branch): in training mode the kernel also accumulates every branch's batch statistics; returns (outputs,
    [per-branch `stats` argument for bn_act, or None])."""
    ks = tuple(int(w.shape[-1]) for w in ws)
    none = [None] * len(ws)
    if x.is_cuda and not T.framework_ops and T.dw_branches_merged and len(ws) > 1 and T._DWB_SETS.get(ks[0]) == ks:
        x = T._autocast(x)
        mult = 8 if x.dtype == torch.float16 else 4
        if T._ok(x, mult):
            bstats = None
            if bns is not None and T.dw_branch_stats and not T._deterministic and all(bn.training and bn.affine for bn in bns):
                bstats = [T.bn_own_scratch(bn, x.device, x.shape[1]) for bn in bns]
            outs = list(T._DWBranches.apply(x, bstats, *ws))
            return (outs, bstats or none) if bns is not None else outs
    outs = [T.dwconv(x, w) for w in ws]
    return (outs, none) if bns is not None else outs


def dwconv(x, w):
    """Depth-wise k x k stride-1 'same' conv (groups == channels) with autograd. w [C,1,k,k], k in {3,5,7,9}."""
    k = w.shape[-1]
    if k == 1:                               # a 1x1 depth-wise conv is a per-channel scale
        return x * w.reshape(1, -1, 1, 1).to(x.dtype)
    if not x.is_cuda or T.framework_ops:       # CPU tensors: plain torch (CI / gloo tests only)
        T.stats["fallback"] += 1
        return F.conv2d(x, w if T.framework_ops else w.to(x.dtype), None, 1, k // 2, 1, x.shape[1])
    x = T._autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (T._ok(x, mult) and k in (3, 5, 7, 9)):
        raise lib.MafError("dwconv: unsupported input for the HIP path: %s %s k=%d" % (tuple(x.shape), x.dtype, k))
    return T._DWConv.apply(x, w)
