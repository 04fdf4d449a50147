"""Training-form dense convolutions on the HIP kernels — 1x1 (forward, data gradient, weight gradient), 3x3 stride 2 (RepVGGBlock.rbr_dense, ConvWrapper), 1x1 stride 2
(RepVGGBlock.rbr_1x1), the two RepVGG convs of the image in one launch — as autograd Functions (yolov6/layers/common.py:29-50, 166-283; engine.py:141-167).

One of the four family files train_ops.py was cut into in round 6 (train_conv / train_dw / train_bn / train_cat).  `T` is train_ops itself: every module-level switch, cache and
helper lives THERE (tests, tools and tape.py read and set them as `train_ops.<name>`), and every reference from here goes through `T.<name>` at call time — so a switch flipped or
an entry point replaced on train_ops (bench.py --torch-convs) reaches this code exactly as it did when all of it was one file.  train_ops re-exports everything defined here;
import train_ops (or the package), not this file."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import lib, pack
from . import train_ops as T


@T._laned
class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, bnslot=None):
        """bnslot: (bn, holder) — the training-mode BatchNorm2d behind the conv and an empty list: when the conv's tile has the statistics epilogue, the
        (scratch, phase) the BatchNorm call must be given as `pre_stats` is appended to the list."""
        x, xs = T.nhwc(x)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dt = T._DT[x.dtype]
        co = -(-cout // 4) * 4                                                   # the kernel stores 4 channels at a time: any class count
        M = B * H * W
        want = bnslot is not None and T.conv_bn_stats and not T._deterministic and bias is None and co == cout
        choice = T._conv_tune.get((M, cin, co, xs, "st") if want else (M, cin, co, xs)) if T.conv_autotune and dt == lib.F16 else None
        w2d = None
        if choice is None:
            w2d = w.detach().reshape(cout, cin).float().contiguous()
            if co != cout:                                                       # (cls_pred with nc % 4 != 0) runs with zero filters appended
                w2d = F.pad(w2d, (0, 0, 0, co - cout))
            choice = T._conv_choice(x, xs, B, H, W, cin, co, dt, w2d, co, cin, 0, want)
        pt, ct, tk = choice
        wp = T._hit(w, ("d", co, cin, 1, 0, dt, ct)) if co == cout else None       # staged by this step's batch (PackPlan)
        if wp is None:
            if w2d is None:
                w2d = w.detach().reshape(cout, cin).float().contiguous()
                if co != cout:
                    w2d = F.pad(w2d, (0, 0, 0, co - cout))
            wp = T._packed_1x1(w2d, co, cin, 0, dt, ct, x.device, w if co == cout else None)
        npad = -(-co // (16 * ct)) * 16 * ct
        if bias is None:
            bp = T._zero_bias(x.device, npad)
        else:
            bp = T._staged_bias(bias, cout, npad, x.device)                       # the bias on the conv's channel tile, zero behind it: staged by the step's pack batch
        out = T._empty((B, co, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        bstat = None
        if bnslot is not None and T._conv_stats_ok(choice, cin, co, cout, dt, bias):
            bstat = T.bn_own_scratch(bnslot[0], x.device, cout)
            bnslot[1].append(bstat)
            T.stats["conv_bn_stats"] = T.stats.get("conv_bn_stats", 0) + 1
        T._launch_conv1x1(x, xs, wp, bp, B, H, W, cin, co, ct, out, dt, pt, tk, bstat)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None   # (an input of this node, not a saved tensor: backward adds its gradient straight into a gradient exchange)
        T.stats["native_conv1x1"] += 1
        return out if co == cout else out[:, :cout]

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = T.nhwc(dy)
        if dy.dtype != x.dtype:
            T._glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = T._DT[x.dtype]
        dx = dw = db = None
        mult = 8 if x.dtype == torch.float16 else 4
        dyk, dyks, kk = dy, dys, cout
        if cout % mult and (ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and x.dtype == torch.float16)):
            kk = -(-cout // mult) * mult                                         # e.g. reg_pred: 68 channels in fp16 — dY zero-padded to whole 16-byte chunks ONCE, for
            if dys >= kk and T.zero_padded.get(dy.data_ptr(), 0) >= kk:            # the weight gradient and the data gradient's reduction dim
                dyk = dy.as_strided((B, kk, H, W), dy.stride())                  # its producer keeps the padding channels zero: no copy
            else:
                T._glue()
                dyk = F.pad(dy, (0, 0, 0, 0, 0, kk - cout)).contiguous(memory_format=torch.channels_last)
                dyks = kk
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ex, view = T._grad_sink(ctx.bias_param) if ctx.bias_param is not None and x.dtype == torch.float16 else (None, None)
            if ex is not None and view.is_contiguous() and cout <= 256:
                # the bias gradient as a column sum on the weight-gradient stream, added into the bias' slice of the gradient exchange (csrc/train_ops.hip
                # maf_colsum) — a framework reduction + an accumulation add on the main stream otherwise
                h = T._fork(x.device, dy)
                lib.check(lib.load().maf_colsum(dy.data_ptr(), dys, B * H * W, cout, dt, view.data_ptr(), h))
                ex.side_done(ctx.bias_param)
                T.stats["native_bias_grad"] = T.stats.get("native_bias_grad", 0) + 1
            else:
                db = dy.sum((0, 2, 3), dtype=torch.float32)
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:                                        # csrc/wgrad.hip: pixel chunks, LDS transpose, MFMA, fp32 atomics
                dw = T._wgrad(x, dyk, dyks, w, 1, 1)                               # any Cin (channel chunks of 256), any Cout (dY padded to 8 channels); None: went into the exchange
            else:                                                                # fp32 parity mode: the framework's TN GEMM
                x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)                      # NHWC rows (a view when x is dense)
                d2 = dy.permute(0, 2, 3, 1).reshape(-1, cout)
                dw = torch.mm(d2.t(), x2).float().reshape(w.shape).to(w.dtype)
                T.stats["framework_wgrad_fp32"] = T.stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            # W^T: dX[m, ci] = sum_co dY[m, co] W[co, ci].  A dY padded to kk > cout channels needs no padded weight: the packer zero-fills K up to whole k-steps,
            # and rounding cout up to 8 never crosses one — the fragment record of [cout] rows IS the one of [kk] rows
            w2d = None
            M = B * H * W
            choice = T._conv_tune.get((M, kk, cin, dyks)) if T.conv_autotune and dt == lib.F16 else None
            if choice is None:
                w2d = w.detach().reshape(cout, cin).float().contiguous()
                choice = T._conv_choice(dyk, dyks, B, H, W, kk, cin, dt, w2d, cout, cin, 1)
            pt, ct, tk = choice
            wp = T._hit(w, ("d", cout, cin, 1, 1, dt, ct))
            if wp is None:
                if w2d is None:
                    w2d = w.detach().reshape(cout, cin).float().contiguous()
                wp = T._packed_1x1(w2d, cout, cin, 1, dt, ct, x.device, w)
            npad = -(-cin // (16 * ct)) * 16 * ct
            dx = T._empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            T._launch_conv1x1(dyk, dyks, wp, T._zero_bias(x.device, npad), B, H, W, kk, cin, ct, dx, dt, pt, tk)
        T._side_done(x.device, dw is not None)
        return dx, dw, db, None


def _wgrad(x, dy, dys, w, ksize, stride):
    """fp16 weight gradient on csrc/wgrad.hip (maf_conv_wgrad): x [B,Cin_x,Hs,Ws], dy [B,Cout,Ho,Wo] NHWC views -> dW like w, fp32 (Cin_x >= w's
    input channels: the stem's image padded to 8).  Launched on the side stream (`_fork`): call it BEFORE the data gradient of the layer is
    launched.  With a gradient exchange the result is accumulated into w's slice of its bucket on that stream and None is returned."""
    B, cin, Hs, Ws = x.shape
    cdy, Ho, Wo = dy.shape[1:]
    cout, cin_w = w.shape[0], w.shape[1]
    xx, xs = T.nhwc(x)
    co = -(-cdy // 8) * 8
    if co != cdy:                                                               # e.g. reg_pred: 68 channels, an odd class count (the 1x1 backward hands dY in padded already)
        T._glue()                                                                 # (torch kernels: not recordable by a step tape)
        dy = F.pad(dy, (0, 0, 0, 0, 0, co - cdy)).contiguous(memory_format=torch.channels_last)
        dys = co
    ex, view = T._grad_sink(w)
    direct = ex is not None and ksize == 1 and co == cout and cin == cin_w      # the kernel's [Cout][Cin] IS the parameter's layout: accumulate in place
    L = lib.load()
    if direct:
        dwf = view
        h = T._fork(x.device, xx, dy)
    else:
        dwf = T._empty((co, cin) if ksize == 1 else (3, 3, co, cin), dtype=torch.float32, device=x.device)       # 3x3: tap-major (csrc/wgrad.hip)
        h = T._fork(x.device, xx, dy, dwf)
        lib.check(L.maf_zero(dwf.data_ptr(), dwf.numel() * 4, h))
    with T._prof("conv_wgrad_k%d" % ksize, (B * Hs * Ws * cin + B * Ho * Wo * co) * 2 + dwf.numel() * 4, x.device, (B, Hs, Ws, cin, co, xs, dys, stride), h):
        lib.check(L.maf_conv_wgrad(xx.data_ptr(), xs, dy.data_ptr(), dys, B, Ho, Wo, Hs, Ws, cin, co, ksize, stride, lib.F16, dwf.data_ptr(), h))
    T.stats["native_wgrad"] = T.stats.get("native_wgrad", 0) + 1
    if ex is not None:
        if not direct:                                                          # tap-major / padded -> the parameter's [Cout][Cin][taps], added on the side stream
            lib.check(L.maf_grad_fold(dwf.data_ptr(), ksize * ksize, co, cin, view.data_ptr(), cout, cin_w, 1, h))
        ex.side_done(w, folded=not direct)
        return None
    if ksize == 3:
        dwf = dwf.permute(2, 3, 0, 1)
    return dwf[:cout, :cin_w].reshape(w.shape).to(w.dtype)


def _tile_dgrad(n, m_pixels):
    """(tile_p, tile_c) for the data-gradient launches: tile_c in {2, 4, 8} (the instantiations of csrc/conv_mfma_dgrad.hip)."""
    ct = 8 if n >= 128 else 4 if n > 32 else 2
    return (2 if -(-m_pixels // 128) * -(-n // (16 * ct)) >= 1024 else 1), ct


def _packed_3x3(w, transpose, dt, ct, dev):
    """Fragment-packed 3x3 weights on the device: tap-major K, every tap padded to whole k-steps == one [N, 9*Kp] matrix in the order of
    maf_pack_w1x1.  transpose: the data gradient's operand (N = the forward conv's input channels, K = its output channels)."""
    cout, cin = w.shape[0], w.shape[1]
    hit = T._hit(w, ("d", cout, cin, 9, int(transpose), dt, ct))
    if hit is not None:
        return hit
    ks = 32 if dt == lib.F16 else 16
    n, k = (cin, cout) if transpose else (cout, cin)
    kp = -(-k // ks) * ks

    def now(dst):
        m = w.detach().float().permute(1, 2, 3, 0) if transpose else w.detach().float().permute(0, 2, 3, 1)      # [N, 3, 3, K]
        big = F.pad(m, (0, kp - k)).reshape(n, 9 * kp).contiguous()
        lib.check(lib.load().maf_pack_w1x1(big.data_ptr(), n, 9 * kp, 0, dt, ct, dst.data_ptr(), T._stream(dev)))

    nbytes = lib.load().maf_pack_w1x1_bytes(n, 9 * kp, 0, dt, ct)
    if not (w.dtype == torch.float32 and w.is_contiguous() and w.is_leaf):        # a temporary (e.g. the stem's channel-padded filters): no plan entry
        buf = T._empty(nbytes, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    fields = dict(kind=0, dtype=dt, Cout=cout, Cin=cin, taps=9, transpose=int(transpose), CT=ct, steps=9 * kp // ks, Kp=kp, flip=0,
                  total=nbytes // (2 if dt == lib.F16 else 4))
    return T._staged(w, ("d", cout, cin, 9, int(transpose), dt, ct), nbytes, fields, now)


def _conv3_choice(op, key, cands, w, transpose, dt, dev):
    """(tile_p, tile_c, tile_k) of a 3 x 3 stride-2 launch (forward: MAF_OP_CONV3X3S2, data gradient: MAF_OP_CONV3X3S2_DGRAD): like `_conv_choice`, every
    candidate is timed once per shape on the tensors at hand (the static rule — pack.tile_for / _tile_dgrad — left the neck's 128 -> 128 side convs at 1.2 TB/s)."""
    best = T._conv3_tune.get(key)
    if best is not None:
        return best
    torch.cuda.synchronize(dev)
    timer, st, res = lib.Timer(), T._stream(dev), []
    saved, T.profile = T.profile, None
    saved_plan, T._plan = T._plan, None                                             # the candidates' weight forms are packed here and now: only the winner's joins the staging plan
    L = lib._lib if lib._lib is not None else lib.load()                        # (never through a recording tape's proxy)
    try:
        for pt, ct, tk in cands:
            n = op.Cout
            op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
            op.w, op.bias = T._packed_3x3(w, transpose, dt, ct, dev).data_ptr(), T._zero_bias(dev, -(-n // (16 * ct)) * 16 * ct).data_ptr()
            if L.maf_op_launch(C.byref(op), st) != 0:
                continue
            ts = []
            for _ in range(3):
                timer.start(st)
                L.maf_op_launch(C.byref(op), st)
                timer.stop(st)
                ts.append(timer.elapsed_ms())
            res.append((min(ts), pt, ct, tk))
    finally:
        T.profile, T._plan = saved, saved_plan
    res.sort()
    best = T._conv3_tune[key] = res[0][1:] if res else cands[-1]
    T.stats["conv_tuned"] = T.stats.get("conv_tuned", 0) + 1
    return best


def _conv3_fwd_cands(cin, cout, M, static):
    cands = []
    ksteps = 9 * -(-cin // 32)
    for ct in (2, 4, 6, 8):
        nt = -(-cout // (16 * ct))
        if nt * 16 * ct > 2 * max(cout, 32) or (ct == 8 and cout % 8):
            continue
        for pt in (1, 2, 4):
            if (pt == 4 and ct > 4) or (pt > 1 and -(-M // (64 * pt)) * nt < 256):
                continue
            cands.append((pt, ct, 1))
        if M <= 65536:
            cands.append((1, ct, 4))                                             # split-K across the four waves
        if ct >= 4:
            for pt in ((1, 2, 4) if ct == 4 else (1, 2)):                        # each k-step's weight fragments through LDS once per workgroup
                if pt == 1 or -(-M // (64 * pt)) * nt >= 256:
                    cands.append((pt, ct, 2))
                    if pt <= 2:
                        cands.append((pt, ct, 8))                                # ... by DMA, two k-steps per barrier (csrc/conv_mfma_dma.hip)
    if static not in cands:
        cands.append(static)
    return cands


def _conv3_dgrad_cands(cin, M, static):
    cands = []
    for ct in (2, 4, 8):
        nt = -(-cin // (16 * ct))
        if nt * 16 * ct > 2 * max(cin, 32):
            continue
        for pt in (1, 2, 4):
            if (pt == 4 and ct > 4) or (pt > 1 and -(-M // (128 * pt)) * nt < 256):
                continue
            cands.append((pt, ct, 0))
    if static not in cands:
        cands.append(static)
    return cands


@T._laned
class _Conv3x3s2(torch.autograd.Function):
    """nn.Conv2d(k=3, stride=2, padding=1, bias=False): forward csrc/conv_mfma.inc.h VAR_3X3S2, data gradient VAR_DGRAD3 (gather form),
    weight gradient csrc/wgrad.hip with the taps gathered in the kernel."""

    @staticmethod
    def forward(ctx, x, w):
        x, xs = T.nhwc(x)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dt = T._DT[x.dtype]
        pt, ct = pack.tile_for(cout, B * Ho * Wo)
        tk = 1
        out = T._empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        op = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV3X3S2, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout, op.nsrc = B, Ho, Wo, H, W, cin, cout, 1
        op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = x.data_ptr(), cin, xs, 0, lib.SRC_DIRECT
        op.out, op.out_stride, op.out_coff = out.data_ptr(), out.stride()[3], 0
        if T.conv3_autotune and dt == lib.F16 and w.dtype == torch.float32 and w.is_leaf:
            key = ("f", B * Ho * Wo, cin, cout, xs)
            ch = T._conv3_tune.get(key)
            if ch is None and T._rec is None:                                      # (never timed inside a recording step: the static tile then)
                ch = T._conv3_choice(op, key, T._conv3_fwd_cands(cin, cout, B * Ho * Wo, (pt, ct, 1)), w, False, dt, x.device)
            if ch is not None:
                pt, ct, tk = ch
        wp = T._packed_3x3(w, False, dt, ct, x.device)
        op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
        op.w, op.bias = wp.data_ptr(), T._zero_bias(x.device, -(-cout // (16 * ct)) * 16 * ct).data_ptr()
        es = x.element_size()
        with T._prof("conv3x3s2", (B * H * W * cin + B * Ho * Wo * cout + 9 * cin * cout) * es, x.device, (B, H, W, cin, cout, pt, ct)):
            lib.check(lib.load().maf_op_launch(C.byref(op), T._stream(x.device)))
        ctx.save_for_backward(x, w)
        T.stats["native_conv3x3s2"] = T.stats.get("native_conv3x3s2", 0) + 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = T.nhwc(dy)
        if dy.dtype != x.dtype:
            T._glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        Ho, Wo = dy.shape[2:]
        dt = T._DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:
                dw = T._wgrad(x, dy, dys, w, 3, 2)
            else:                                                                # fp32 parity mode: the framework's kernel
                dw = torch.nn.grad.conv2d_weight(x[:, :w.shape[1]], w.shape, dy, stride=2, padding=1).to(w.dtype)
                T.stats["framework_wgrad_fp32"] = T.stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            # (an input whose channels were padded — the image — gets zeros in the padding: the packer pads W^T's rows to the channel tile)
            pt, ct = T._tile_dgrad(cin, B * H * W)
            dx = T._empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            op = lib.MafOp()
            op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV3X3S2_DGRAD, dt, dt, lib.ACT_NONE
            op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout, op.nsrc = B, H, W, Ho, Wo, cout, cin, 1
            op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = dy.data_ptr(), cout, dys, 0, lib.SRC_DIRECT
            op.out, op.out_stride, op.out_coff = dx.data_ptr(), dx.stride()[3], 0
            if T.conv3_autotune and dt == lib.F16 and w.dtype == torch.float32 and w.is_leaf:
                key = ("d", B * H * W, cin, cout, dys)
                ch = T._conv3_tune.get(key)
                if ch is None and T._rec is None:
                    ch = T._conv3_choice(op, key, T._conv3_dgrad_cands(cin, B * H * W, (pt, ct, 0)), w, True, dt, x.device)
                if ch is not None:
                    pt, ct = ch[0], ch[1]
            wp = T._packed_3x3(w, True, dt, ct, x.device)
            op.tile_p, op.tile_c, op.tile_k = pt, ct, 0
            op.w, op.bias = wp.data_ptr(), T._zero_bias(x.device, -(-cin // (16 * ct)) * 16 * ct).data_ptr()
            es = x.element_size()
            with T._prof("conv3x3s2_dgrad", (B * H * W * cin + B * Ho * Wo * cout + 9 * cin * cout) * es, x.device, (B, H, W, cin, cout, pt, ct)):
                lib.check(lib.load().maf_op_launch(C.byref(op), T._stream(x.device)))
        T._side_done(x.device, dw is not None)
        return dx, dw


@T._laned
class _Conv1x1s2(torch.autograd.Function):
    """nn.Conv2d(k=1, stride=2, bias=False) (RepVGGBlock.rbr_1x1, common.py:203): the 1x1 kernel reading pixel (2y, 2x) of its source
    (MAF_SRC_SUB2); data gradient = the 1x1 data gradient scattered onto the even pixels; weight gradient csrc/wgrad.hip, one gathered tap."""

    @staticmethod
    def forward(ctx, x, w):
        x, xs = T.nhwc(x)
        B, cin, H, W = x.shape
        assert H % 2 == 0 and W % 2 == 0, "stride-2 1x1 conv: even input sides (images are multiples of 32)"
        cout = w.shape[0]
        Ho, Wo = H // 2, W // 2
        dt = T._DT[x.dtype]
        pt, ct = pack.tile_for(cout, B * Ho * Wo)
        cin_w = w.shape[1]                                                       # < cin for the stem: the image's channels are padded to 8, the weight's K to whole k-steps by the packer
        wp = T._packed_1x1(w.detach().reshape(cout, cin_w).float().contiguous(), cout, cin_w, 0, dt, ct, x.device, w)
        out = T._empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        op = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV1X1, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Cin, op.Cout, op.nsrc = B, Ho, Wo, cin, cout, 1
        op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = x.data_ptr(), cin, xs, 0, lib.SRC_SUB2
        op.out, op.out_stride, op.out_coff = out.data_ptr(), out.stride()[3], 0
        op.tile_p, op.tile_c = pt, ct
        op.w, op.bias = wp.data_ptr(), T._zero_bias(x.device, -(-cout // (16 * ct)) * 16 * ct).data_ptr()
        with T._prof("conv1x1", B * Ho * Wo * (cin + cout) * x.element_size(), x.device):
            lib.check(lib.load().maf_op_launch(C.byref(op), T._stream(x.device)))
        ctx.save_for_backward(x, w)
        T.stats["native_conv1x1"] += 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = T.nhwc(dy)
        if dy.dtype != x.dtype:
            T._glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = T._DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:
                dw = T._wgrad(x, dy, dys, w, 1, 2)
            else:
                xsub = x[:, :w.shape[1], ::2, ::2].permute(0, 2, 3, 1).reshape(-1, w.shape[1])
                dw = torch.mm(dy.permute(0, 2, 3, 1).reshape(-1, cout).t(), xsub).float().reshape(w.shape).to(w.dtype)
                T.stats["framework_wgrad_fp32"] = T.stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            cin_w = w.shape[1]                                                   # < cin: padded image channels get a zero gradient (W^T's rows are padded to the channel tile)
            ct = pack.tile_for(cin, B * (H // 2) * (W // 2))[1]
            wp = T._packed_1x1(w.detach().reshape(cout, cin_w).float().contiguous(), cout, cin_w, 1, dt, ct, x.device, w)
            dxs = T._empty((B, cin, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            T._launch_conv1x1(dy, dys, wp, T._zero_bias(x.device, -(-cin // (16 * ct)) * 16 * ct), B, H // 2, W // 2, cout, cin, ct, dxs, dt)
            if getattr(ctx, "compact", False):                                   # _RepVGGConvs adds it onto the 3x3 branch's data gradient itself
                dx = dxs
            else:
                T._glue(2)
                dx = T._empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
                dx[:, :, ::2, ::2] = dxs
        T._side_done(x.device, dw is not None)
        return dx, dw


class _Ctx:
    """What a Function's forward / backward use of their ctx, for calling them from another Function."""
    needs_input_grad = (True, True)

    def save_for_backward(self, *t):
        self.saved_tensors = t


@T._laned
class _RepVGGConvs(torch.autograd.Function):
    """(conv3x3 s2 (x, w3), conv1x1 s2 (x, w1)) — the two branches of a RepVGGBlock (yolov6/layers/common.py:199-203) as ONE autograd node, so that their
    data gradients meet inside it: the 1x1 branch's gradient lives on the even pixels only and is added onto the 3x3 branch's in place (maf_add_sub2, a quarter
    of the pixels) instead of a zero-filled full-size tensor + a strided copy + autograd's full-size add."""

    @staticmethod
    def forward(ctx, x, w3, w1):
        if T.stem_train and x.dtype == torch.float16 and w3.shape[1] == 3 and x.shape[1] == 8 and w3.shape[0] % 8 == 0 and w3.shape[0] <= 96 \
                and w3.dtype == torch.float32 and w1.dtype == torch.float32 and w3.is_contiguous() and w1.is_contiguous() and x.shape[3] % 8 == 0:
            # the image (3 channels padded to 8): both branches in ONE launch of a direct conv (csrc/stem_train.hip) — the generic kernels read it twice with a K of 72 / 8
            x, xs = T.nhwc(x)
            B, _, H, W = x.shape
            cout = w3.shape[0]
            z3 = T._empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            z1 = T._empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            with T._prof("stem_train", (B * H * W * 8 + 2 * B * (H // 2) * (W // 2) * cout) * 2, x.device, (B, H, W, cout)):
                lib.check(lib.load().maf_stem_train(x.data_ptr(), xs, B, H, W, w3.data_ptr(), w1.data_ptr(), cout, z3.data_ptr(), z1.data_ptr(), T._stream(x.device)))
            ctx.save_for_backward(x, w3, w1)
            T.stats["native_conv3x3s2"] = T.stats.get("native_conv3x3s2", 0) + 1
            T.stats["native_conv1x1"] += 1
            T.stats["native_stem_train"] = T.stats.get("native_stem_train", 0) + 1
            return z3, z1
        c3, c1 = T._Ctx(), T._Ctx()
        z3 = T._Conv3x3s2.forward(c3, x, w3)
        z1 = T._Conv1x1s2.forward(c1, x, w1)
        ctx.save_for_backward(c3.saved_tensors[0], w3, w1)
        return z3, z1

    @staticmethod
    def backward(ctx, dz3, dz1):
        x, w3, w1 = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        c3, c1 = T._Ctx(), T._Ctx()
        c3.saved_tensors, c3.needs_input_grad = (x, w3), (need_x, ctx.needs_input_grad[1])
        c1.saved_tensors, c1.needs_input_grad, c1.compact = (x, w1), (need_x, ctx.needs_input_grad[2]), True
        dx, dw3 = T._Conv3x3s2.backward(c3, dz3)
        dxs, dw1 = T._Conv1x1s2.backward(c1, dz1)
        if need_x:
            B, c, Ho, Wo = dxs.shape
            lib.check(lib.load().maf_add_sub2(dxs.data_ptr(), dxs.stride()[3], dx.data_ptr(), dx.stride()[3], B, Ho, Wo, c, T._DT[dx.dtype], T._stream(dx.device)))
        return dx, dw3, dw1


def repvgg_convs(x, w3, w1):
    """(conv3x3s2(x, w3), conv1x1s2(x, w1)) of one input."""
    if not x.is_cuda or T.framework_ops or x.shape[2] % 2 or x.shape[3] % 2:
        return T.conv3x3s2(x, w3), T.conv1x1s2(x, w1)
    x = T._autocast(x)
    if not (x.dtype in T._DT and x.dim() == 4 and tuple(w3.shape[2:]) == (3, 3) and tuple(w1.shape[2:]) == (1, 1)):
        raise lib.MafError("repvgg_convs: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return T._RepVGGConvs.apply(T._pad8(x, w3), w3, w1)


def pad_channels8(x):
    """A 3-channel image for kernels that read 16-byte channel chunks: cast as a convolution would under autocast, zero channels appended.
    RepVGGBlock does it ONCE for its two branches (32 x 3 x 640 x 640: the cast and the padded copy cost 0.15 ms each)."""
    x = T._autocast(x)
    cin = x.shape[1]
    if cin % 8 == 0 or not x.is_cuda:
        return x
    return F.pad(x, (0, 0, 0, 0, 0, 8 - cin % 8)).contiguous(memory_format=torch.channels_last)


def _pad8(x, w):
    """x with its channels padded to a multiple of 8 (it may come padded already: pad_channels8).  The WEIGHT stays the parameter itself:
    the packers zero-pad its K to whole k-steps, and the weight-gradient path slices the valid input channels back out (`_wgrad`), so the
    parameter's gradient never passes through an autograd pad node on the main stream."""
    cin = w.shape[1]
    if x.shape[1] == cin:
        if cin % 8 == 0:
            return x
        x = T.pad_channels8(x)
    if x.shape[1] != -(-cin // 8) * 8:
        raise lib.MafError("conv: input has %d channels, the filters %d" % (x.shape[1], cin))
    return x


def conv3x3s2(x, w):
    """nn.Conv2d(k=3, stride=2, padding=1, bias=False) with autograd; x [B,Cin,H,W] (NHWC in memory preferred), w [Cout,Cin,3,3]."""
    if not x.is_cuda or T.framework_ops:      # CPU tensors: the train-form module tree in plain torch (CI / gloo tests only)
        T.stats["fallback"] += 1
        if T.framework_ops and x.shape[1] > w.shape[1]:      # RepVGGBlock hands the image zero-padded to 8 channels (pad_channels8)
            x = x[:, :w.shape[1]]
        return F.conv2d(x, w if T.framework_ops else w.to(x.dtype), None, 2, 1)
    x = T._autocast(x)
    if not (x.dtype in T._DT and x.dim() == 4 and tuple(w.shape[2:]) == (3, 3)):
        raise lib.MafError("conv3x3s2: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return T._Conv3x3s2.apply(T._pad8(x, w), w)


def conv1x1s2(x, w):
    """nn.Conv2d(k=1, stride=2, bias=False) with autograd."""
    if not x.is_cuda or T.framework_ops:
        T.stats["fallback"] += 1
        if T.framework_ops and x.shape[1] > w.shape[1]:
            x = x[:, :w.shape[1]]
        return F.conv2d(x, w if T.framework_ops else w.to(x.dtype), None, 2, 0)
    x = T._autocast(x)
    if not (x.dtype in T._DT and x.dim() == 4 and tuple(w.shape[2:]) == (1, 1)):
        raise lib.MafError("conv1x1s2: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return T._Conv1x1s2.apply(T._pad8(x, w), w)


def conv1x1_bn(x, w, bn):
    """(conv1x1(x, w), pre_stats): the 1x1 conv in front of the BatchNorm2d `bn` (Conv.forward, yolov6/layers/common.py:44-47).  When `bn` normalises with batch
    statistics on the HIP path and the conv runs on the persistent LDS-weight kernel, the conv's epilogue accumulates them and `pre_stats` is what
    bn_act(..., pre_stats=) takes (apply pass only); otherwise None."""
    if not (x.is_cuda and bn.training and bn.affine) or T.framework_ops or not T.conv_bn_stats or T._deterministic:
        return T.conv1x1(x, w), None
    x = T._autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (T._ok(x, mult) and w.shape[2] == 1):
        raise lib.MafError("conv1x1: unsupported input for the HIP path: %s %s -> %d channels" % (tuple(x.shape), x.dtype, w.shape[0]))
    holder = []
    z = T._Conv1x1.apply(x, w, None, (bn, holder))
    return z, (holder[0] if holder else None)


def conv1x1(x, w, bias=None):
    """nn.Conv2d(k=1, stride=1) forward with autograd. x [B,Cin,H,W] (NHWC in memory preferred), w [Cout,Cin,1,1]."""
    if not x.is_cuda or T.framework_ops:      # CPU tensors: the train-form module tree in plain torch (CI / gloo tests only)
        T.stats["fallback"] += 1
        if T.framework_ops:                   # autocast (if any) casts the operands itself
            return F.conv2d(x, w, bias)
        return F.conv2d(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))
    x = T._autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (T._ok(x, mult) and w.shape[2] == 1):
        raise lib.MafError("conv1x1: unsupported input for the HIP path: %s %s -> %d channels" % (tuple(x.shape), x.dtype, w.shape[0]))
    return T._Conv1x1.apply(x, w, bias)
