"""-m gpu: per-kernel parity of the HIP ops, called through the C-ABI (maf_op_launch), against plain
PyTorch fp32 references of the same op on the CPU.

Tolerances: f32 kernels 2e-5 rel-to-max (fp32 accumulation-order noise); f16 kernels: inputs and
weights are rounded to fp16 first so only the fp16 store rounding + accumulation order remain:
|d| <= 2e-3*max|ref| + 2e-3.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from maf_yolo_amd import lib, pack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {lib.F16: torch.float16, lib.F32: torch.float32}


def _launch(op):
    lib.check(lib.load().maf_op_launch(C.byref(op), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()


def _nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(DT[dt]).to(DEV)


def _check(out_nhwc, ref_nchw, dt):
    got = out_nhwc.float().cpu().permute(0, 3, 1, 2)
    scale = ref_nchw.abs().max().item()
    tol = 2e-5 * scale + 1e-6 if dt == lib.F32 else 2e-3 * scale + 2e-3
    err = (got - ref_nchw).abs().max().item()
    assert err <= tol, (err, tol)


def _q(t, dt):
    return t.half().float() if dt == lib.F16 else t


def _act(y, act):
    return {lib.ACT_NONE: y, lib.ACT_RELU: F.relu(y), lib.ACT_SILU: F.silu(y), lib.ACT_SIGMOID: torch.sigmoid(y)}[act]


def _conv_op(kind, dt, B, H, W, cin, cout, act, srcs, out, out_stride, out_coff, w, b, pt, ct, Hin=0, Win=0, out_f32=0):
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = kind, dt, dt, act
    op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout = B, H, W, Hin, Win, cin, cout
    op.nsrc = len(srcs)
    for i, (t, c, stride, coff, mode) in enumerate(srcs):
        op.src[i].ptr, op.src[i].C, op.src[i].stride, op.src[i].coff, op.src[i].mode = t.data_ptr(), c, stride, coff, mode
    op.out, op.out_stride, op.out_coff, op.out_f32 = out.data_ptr(), out_stride, out_coff, out_f32
    op.tile_p, op.tile_c = pt, ct
    op.w, op.bias = w.data_ptr(), b.data_ptr()
    return op


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("cin,cout,pt,ct,act", [(48, 48, 2, 4, lib.ACT_SILU), (72, 24, 1, 2, lib.ACT_SILU), (24, 72, 2, 6, lib.ACT_NONE),
                                                (192, 128, 1, 8, lib.ACT_SILU), (128, 80, 2, 6, lib.ACT_SIGMOID), (64, 64, 1, 4, lib.ACT_RELU),
                                                (128, 68, 1, 6, lib.ACT_NONE), (40, 200, 2, 4, lib.ACT_SILU)])
def test_conv1x1_direct(dt, cin, cout, pt, ct, act):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    B, H, W = 2, 9, 13                                   # 234 pixels: ragged last tile
    x = _q(torch.randn(B, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    b = torch.randn(cout, generator=g)
    ref = _act(F.conv2d(x, w, b), act)
    xs = _nhwc(x, dt)
    wp = pack.pack_conv1x1(w, [cin], ct, dt).to(DEV)
    bp = pack.pack_bias(b, ct).to(DEV)
    # write into a channel slice of a wider buffer (concat-slice epilogue)
    stride, coff = cout + 16, 8
    out = torch.full((B, H, W, stride), 7.0, dtype=DT[dt], device=DEV)
    _launch(_conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, act, [(xs, cin, cin, 0, 0)], out, stride, coff, wp, bp, pt, ct))
    _check(out[..., coff:coff + cout], ref, dt)
    assert (out[..., :coff] == 7).all() and (out[..., coff + cout:] == 7).all(), "wrote outside its slice"


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("kind", ["1x1", "3x3"])
def test_conv_split_k(dt, kind):
    """tile_k = 4: the four waves of a workgroup split the reduction (small maps, long K)."""
    g = torch.Generator().manual_seed(9)
    B, cin, cout, ct = 2, 384, 192, 6
    if kind == "1x1":
        H = W = 7
        x = _q(torch.randn(B, cin, H, W, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [cin], ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(x, w, b))
        opk, Hin, Win = lib.OP_CONV1X1, 0, 0
    else:
        cin = 96
        Hin, Win = 10, 14
        H, W = 5, 7
        x = _q(torch.randn(B, cin, Hin, Win, generator=g), dt)
        w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), dt)
        wp = pack.pack_conv3x3(w, ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(x, w, b, 2, 1))
        opk = lib.OP_CONV3X3S2
    bias = torch.randn(cout, generator=g)
    out = torch.zeros(B, H, W, cout, dtype=DT[dt], device=DEV)
    op = _conv_op(opk, dt, B, H, W, cin, cout, lib.ACT_SILU, [(_nhwc(x, dt), cin, cin, 0, 0)], out, cout, 0, wp.to(DEV), pack.pack_bias(bias, ct).to(DEV), 1, ct, Hin=Hin, Win=Win)
    op.tile_k = 4
    _launch(op)
    _check(out, ref_fn(bias), dt)


@pytest.mark.parametrize("tk", [2, 8])
@pytest.mark.parametrize("kind,pt,ct", [("1x1", 2, 8), ("1x1", 1, 6), ("1x1", 2, 4), ("3x3", 2, 4), ("3x3", 1, 8), ("3x3", 2, 6), ("3x3", 4, 4), ("multi", 2, 6), ("multi", 1, 4), ("multi3", 2, 8)])
def test_conv_lds_shared_weights(kind, pt, ct, tk):
    """tile_k = 2: the workgroup stages each k-step's weight fragments in LDS once for its four waves (long reductions).  tile_k = 8: the fragments arrive by DMA,
    in stages of two k-steps through a ring of four slots (odd step counts, fewer stages than slots, three concatenated sources with an up-sampled one in the middle)."""
    if tk == 8 and pt > 2:
        pytest.skip("tile_k = 8 has tile_p = 1 / 2")
    g = torch.Generator().manual_seed(17 + pt + ct)
    dt, B = lib.F16, 2
    cout = {8: 128, 6: 88, 4: 64}[ct]
    Hin = Win = 0
    if kind == "1x1":
        cin, H, W = 200, 11, 13
        x = _q(torch.randn(B, cin, H, W, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [cin], ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(x, w, b))
        opk, srcs = lib.OP_CONV1X1, [(_nhwc(x, dt), cin, cin, 0, 0)]
    elif kind == "3x3":
        cin, Hin, Win, H, W = 72, 18, 22, 9, 11
        x = _q(torch.randn(B, cin, Hin, Win, generator=g), dt)
        w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), dt)
        wp = pack.pack_conv3x3(w, ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(x, w, b, 2, 1))
        opk, srcs = lib.OP_CONV3X3S2, [(_nhwc(x, dt), cin, cin, 0, 0)]
    elif kind == "multi3":
        ca, cb, cc, H, W = 104, 72, 256, 10, 14
        cin = ca + cb + cc
        a = _q(torch.randn(B, ca, H, W, generator=g), dt)
        bsm = _q(torch.randn(B, cb, H // 2, W // 2, generator=g), dt)
        c3 = _q(torch.randn(B, cc, H, W, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [ca, cb, cc], ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(torch.cat([a, F.interpolate(bsm, scale_factor=2, mode="nearest"), c3], 1), w, b))
        opk, srcs = lib.OP_CONV1X1, [(_nhwc(a, dt), ca, ca, 0, 0), (_nhwc(bsm, dt), cb, cb, 0, 1), (_nhwc(c3, dt), cc, cc, 0, 0)]
    else:
        ca, cb, H, W = 40, 64, 8, 12
        cin = ca + cb
        a = _q(torch.randn(B, ca, H, W, generator=g), dt)
        bsm = _q(torch.randn(B, cb, H // 2, W // 2, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [ca, cb], ct, dt)
        ref_fn = lambda b: F.silu(F.conv2d(torch.cat([a, F.interpolate(bsm, scale_factor=2, mode="nearest")], 1), w, b))
        opk, srcs = lib.OP_CONV1X1, [(_nhwc(a, dt), ca, ca, 0, 0), (_nhwc(bsm, dt), cb, cb, 0, 1)]
    bias = torch.randn(cout, generator=g)
    out = torch.zeros(B, H, W, cout + 8, dtype=DT[dt], device=DEV)
    op = _conv_op(opk, dt, B, H, W, cin, cout, lib.ACT_SILU, srcs, out, cout + 8, 8, wp.to(DEV), pack.pack_bias(bias, ct).to(DEV), pt, ct, Hin=Hin, Win=Win)
    op.tile_k = tk
    _launch(op)
    _check(out[..., 8:], ref_fn(bias), dt)
    assert (out[..., :8] == 0).all()


@pytest.mark.parametrize("cin,cout,pt,ct,act", [(48, 48, 2, 4, lib.ACT_SILU), (24, 72, 2, 6, lib.ACT_SILU), (128, 64, 1, 4, lib.ACT_RELU), (64, 128, 2, 8, lib.ACT_SILU),
                                                (72, 24, 1, 2, lib.ACT_NONE), (96, 200, 2, 4, lib.ACT_SILU), (40, 28, 2, 2, lib.ACT_SIGMOID)])
def test_conv1x1_persistent_stream(cin, cout, pt, ct, act):
    """tile_k = 3: persistent waves, weights resident in registers, next tile's activations prefetched across the epilogue."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    B, H, W, dt = 3, 37, 41, lib.F16                     # 4551 pixels: many tiles per wave would need a big map; ragged tail here
    x = _q(torch.randn(B, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    b = torch.randn(cout, generator=g)
    ref = _act(F.conv2d(x, w, b), act)
    xs = torch.zeros(B, H, W, cin + 8, dtype=torch.float16, device=DEV)
    xs[..., 8:] = _nhwc(x, dt)
    stride, coff = cout + 16, 8
    out = torch.full((B, H, W, stride), 7.0, dtype=DT[dt], device=DEV)
    op = _conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, act, [(xs, cin, cin + 8, 8, 0)], out, stride, coff,
                  pack.pack_conv1x1(w, [cin], ct, dt).to(DEV), pack.pack_bias(b, ct).to(DEV), pt, ct)
    op.tile_k = 3
    _launch(op)
    _check(out[..., coff:coff + cout], ref, dt)
    assert (out[..., :coff] == 7).all() and (out[..., coff + cout:] == 7).all(), "wrote outside its slice"


@pytest.mark.parametrize("kind,cin,cout,ct", [("direct", 200, 128, 8), ("direct", 384, 96, 6), ("direct", 72, 48, 4), ("direct", 144, 24, 2), ("direct", 576, 192, 6), ("direct", 768, 96, 6), ("direct", 576, 128, 8), ("direct", 448, 64, 4),
                                              ("multi", 0, 96, 6), ("multi", 0, 128, 8), ("multi", 0, 64, 4), ("multi8", 0, 128, 8), ("multi8", 0, 192, 6),
                                              # round 6 (csrc/conv_stream_lds_xwide.hip): 26 .. 40 k-steps at tile_c = 4 — the 832 ... 1280-channel reductions of s / m
                                              ("direct", 1024, 128, 4), ("direct", 1280, 64, 4), ("direct", 896, 192, 4), ("direct", 1152, 80, 4), ("multi26", 0, 192, 4), ("multi34", 0, 64, 4)])
def test_conv1x1_persistent_lds_weights(kind, cin, cout, ct):
    """tile_k = 5: persistent waves with the channel tile's weights resident in LDS; single source or concat (with upsample)."""
    g = torch.Generator().manual_seed(31 + cout + ct)
    dt, B, H, W = lib.F16, 2, 18, 22
    if kind == "direct":
        x = _q(torch.randn(B, cin, H, W, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [cin], ct, dt)
        xs = torch.zeros(B, H, W, cin + 8, dtype=torch.float16, device=DEV)
        xs[..., 8:] = _nhwc(x, dt)
        srcs, full = [(xs, cin, cin + 8, 8, 0)], x
    else:
        ca, cb, cc = {"multi": (40, 64, 24), "multi8": (128, 192, 128), "multi26": (320, 256, 256), "multi34": (384, 512, 192)}[kind]   # multi8: a wide MAFPN concat (14 k-steps), the shapes the eight-wave form is for; multi26 / 34: s / m (backbone.26.conv1, backbone.16.conv1)
        cin = ca + cb + cc
        a = _q(torch.randn(B, ca, H, W, generator=g), dt)
        bsm = _q(torch.randn(B, cb, H // 2, W // 2, generator=g), dt)
        c = _q(torch.randn(B, cc, H, W, generator=g), dt)
        w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
        wp = pack.pack_conv1x1(w, [ca, cb, cc], ct, dt)
        srcs = [(_nhwc(a, dt), ca, ca, 0, 0), (_nhwc(bsm, dt), cb, cb, 0, 1), (_nhwc(c, dt), cc, cc, 0, 0)]
        full = torch.cat([a, F.interpolate(bsm, scale_factor=2, mode="nearest"), c], 1)
    bias = torch.randn(cout, generator=g)
    ref = F.silu(F.conv2d(full, w, bias))
    out = torch.full((B, H, W, cout + 8), 5.0, dtype=DT[dt], device=DEV)
    op = _conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, lib.ACT_SILU, srcs, out, cout + 8, 8, wp.to(DEV), pack.pack_bias(bias, ct).to(DEV), 1, ct)
    op.tile_k = 5
    _launch(op)
    _check(out[..., 8:], ref, dt)
    assert (out[..., :8] == 5).all()
    # tile_p = 2: eight waves per workgroup behind one LDS copy of the weights (csrc/conv_stream_lds_w8.hip; 64 <= k-steps x tile_c <= 160): the same tiles, the
    # same order of additions — bit-identical; a clean error where the instantiation does not exist
    ksteps = sum(-(-s_[1] // 32) for s_ in srcs)
    out8 = torch.full((B, H, W, cout + 8), 5.0, dtype=DT[dt], device=DEV)
    op8 = _conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, lib.ACT_SILU, srcs, out8, cout + 8, 8, wp.to(DEV), pack.pack_bias(bias, ct).to(DEV), 2, ct)
    op8.tile_k = 5
    if ct >= 4 and 64 <= ksteps * ct <= 160 and 8 <= ksteps <= 24:
        _launch(op8)
        assert torch.equal(out8, out)
    else:
        assert lib.load().maf_op_launch(C.byref(op8), torch.cuda.current_stream().cuda_stream) != 0
    # the same conv storing PIXEL PAIRS (out_pairs: the layout csrc/dwconv_p2.hip reads): the same values, [B, H, W/2, stride, 2], a slice of a wider pair buffer
    outp = torch.full((B, H, W // 2, (cout + 8) * 2), 5.0, dtype=DT[dt], device=DEV)
    op2 = _conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, lib.ACT_SILU, srcs, outp, cout + 8, 8, wp.to(DEV), pack.pack_bias(bias, ct).to(DEV), 1, ct)
    op2.tile_k, op2.out_pairs = 5, 1
    _launch(op2)
    got, want = outp[..., 16:].reshape(B, H, W // 2, cout, 2).float(), pack.pairs_from_nhwc(out[..., 8:].contiguous()).float()
    # the same accumulators through a differently scheduled epilogue: equal up to one fp16 rounding of the activation on a handful of values
    assert (got - want).abs().max() <= 1e-3 * want.abs().max() and (got != want).float().mean() < 1e-3
    assert (outp[..., :16] == 5).all()
    op2.tile_k = 1
    assert lib.load().maf_op_launch(C.byref(op2), torch.cuda.current_stream().cuda_stream) != 0       # only the tile_k = 5 epilogue knows the layout


def test_conv1x1_out_f32():
    g = torch.Generator().manual_seed(5)
    B, H, W, cin, cout = 1, 8, 8, 128, 68
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().float()
    b = torch.randn(cout, generator=g)
    out = torch.zeros(B, H, W, cout, dtype=torch.float32, device=DEV)
    _launch(_conv_op(lib.OP_CONV1X1, lib.F16, B, H, W, cin, cout, 0, [(_nhwc(x, lib.F16), cin, cin, 0, 0)], out, cout, 0,
                     pack.pack_conv1x1(w, [cin], 6, lib.F16).to(DEV), pack.pack_bias(b, 6).to(DEV), 1, 6, out_f32=1))
    _check(out, F.conv2d(x, w, b), lib.F32)              # fp32 store: only accumulation-order noise remains


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
def test_conv1x1_multi_source_with_upsample(dt):
    """cat[a (direct, slice of a wider buffer), up2(b), c] -> 1x1: MAF-YOLO-n.yaml:22-23 pattern."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 8, 12
    ca, cb, cc, cout = 24, 64, 40, 96
    a = _q(torch.randn(B, ca, H, W, generator=g), dt)
    bsm = _q(torch.randn(B, cb, H // 2, W // 2, generator=g), dt)
    c = _q(torch.randn(B, cc, H, W, generator=g), dt)
    w = _q(torch.randn(cout, ca + cb + cc, 1, 1, generator=g) / 11.0, dt)
    bias = torch.randn(cout, generator=g)
    ref = F.silu(F.conv2d(torch.cat([a, F.interpolate(bsm, scale_factor=2, mode="nearest"), c], 1), w, bias))
    abuf = torch.zeros(B, H, W, ca + 16, dtype=DT[dt], device=DEV)
    abuf[..., 8:8 + ca] = _nhwc(a, dt)
    out = torch.zeros(B, H, W, cout, dtype=DT[dt], device=DEV)
    srcs = [(abuf, ca, ca + 16, 8, lib.SRC_DIRECT), (_nhwc(bsm, dt), cb, cb, 0, lib.SRC_UP2), (_nhwc(c, dt), cc, cc, 0, lib.SRC_DIRECT)]
    _launch(_conv_op(lib.OP_CONV1X1, dt, B, H, W, ca + cb + cc, cout, lib.ACT_SILU, srcs, out, cout, 0,
                     pack.pack_conv1x1(w, [ca, cb, cc], 6, dt).to(DEV), pack.pack_bias(bias, 6).to(DEV), 1, 6))
    _check(out, ref, dt)


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
def test_conv1x1_maxpool_source(dt):
    """Conv1x1(MaxPool2d(2)(x)) — MPRep.conv1(mp(x)) (common.py:787-788)."""
    g = torch.Generator().manual_seed(12)
    B, H, W, cin, cout = 2, 6, 10, 48, 48
    x = _q(torch.randn(B, cin, 2 * H, 2 * W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 1, 1, generator=g) / 7.0, dt)
    bias = torch.randn(cout, generator=g)
    ref = F.silu(F.conv2d(F.max_pool2d(x, 2, 2), w, bias))
    out = torch.zeros(B, H, W, cout, dtype=DT[dt], device=DEV)
    _launch(_conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, lib.ACT_SILU, [(_nhwc(x, dt), cin, cin, 0, lib.SRC_POOL2)], out, cout, 0,
                     pack.pack_conv1x1(w, [cin], 4, dt).to(DEV), pack.pack_bias(bias, 4).to(DEV), 1, 4))
    _check(out, ref, dt)
    if dt == lib.F16:
        # the same layer on the persistent kernel with the weights in LDS (tile_k = 5): pooled and sub-sampled (1x1 stride 2) single sources, a slice of a wider buffer
        for ct in (2, 4, 6):
            for mode, xin in ((lib.SRC_POOL2, F.max_pool2d(x, 2, 2)), (lib.SRC_SUB2, x[:, :, ::2, ::2])):
                xs = torch.zeros(B, 2 * H, 2 * W, cin + 16, dtype=torch.float16, device=DEV)
                xs[..., 8:8 + cin] = _nhwc(x, dt)
                out5 = torch.full((B, H, W, cout + 8), 5.0, dtype=torch.float16, device=DEV)
                op = _conv_op(lib.OP_CONV1X1, dt, B, H, W, cin, cout, lib.ACT_SILU, [(xs, cin, cin + 16, 8, mode)], out5, cout + 8, 8,
                              pack.pack_conv1x1(w, [cin], ct, dt).to(DEV), pack.pack_bias(bias, ct).to(DEV), 1, ct)
                op.tile_k = 5
                _launch(op)
                _check(out5[..., 8:], F.silu(F.conv2d(xin, w, bias)), dt)
                assert (out5[..., :8] == 5).all()


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("cin,cout,pt,ct,act", [(24, 48, 2, 4, lib.ACT_RELU), (48, 64, 1, 4, lib.ACT_SILU), (128, 128, 1, 8, lib.ACT_SILU), (96, 96, 2, 6, lib.ACT_RELU)])
def test_conv3x3_stride2(dt, cin, cout, pt, ct, act):
    g = torch.Generator().manual_seed(cin + cout)
    B, Hin, Win = 2, 12, 20
    x = _q(torch.randn(B, cin, Hin, Win, generator=g), dt)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), dt)
    bias = torch.randn(cout, generator=g)
    ref = _act(F.conv2d(x, w, bias, 2, 1), act)
    H, W = Hin // 2, Win // 2
    out = torch.zeros(B, H, W, 2 * cout, dtype=DT[dt], device=DEV)
    _launch(_conv_op(lib.OP_CONV3X3S2, dt, B, H, W, cin, cout, act, [(_nhwc(x, dt), cin, cin, 0, 0)], out, 2 * cout, cout,
                     pack.pack_conv3x3(w, ct, dt).to(DEV), pack.pack_bias(bias, ct).to(DEV), pt, ct, Hin=Hin, Win=Win))
    _check(out[..., cout:], ref, dt)
    assert (out[..., :cout] == 0).all()


@pytest.mark.parametrize("cin,cout,act,hw", [(128, 128, lib.ACT_SILU, (40, 40)), (128, 128, lib.ACT_RELU, (21, 35)), (96, 96, lib.ACT_RELU, (38, 50)), (96, 64, lib.ACT_SILU, (80, 80)),
                                             (64, 64, lib.ACT_SILU, (16, 18)), (128, 128, lib.ACT_SILU, (80, 80)),
                                             (64, 96, lib.ACT_SILU, (40, 44)), (64, 96, lib.ACT_SILU, (13, 22)), (128, 96, lib.ACT_SILU, (41, 40))])   # round 6: the side convs of s
@pytest.mark.parametrize("twin", [False, True])
def test_conv3x3_stride2_register_resident_weights(cin, cout, act, hw, twin):
    """tile_k = 7 (csrc/conv3s2_wreg.hip): every weight fragment in registers, the input patch of a 4 x 8 output tile by DMA into a source-permuted
    (bank-conflict-free) LDS image, double-buffered; odd input sizes, tiles hanging over the map, channel slices on both sides, one and two convs
    per launch, any workgroup count; against F.conv2d in fp32 on the same fp16 operands, and equal to the generic MFMA template to summation-order noise."""
    g = torch.Generator().manual_seed(cin * 5 + cout + hw[0])
    B, (Hin, Win) = 3, hw
    H, W = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
    convs = []
    for k in range(2 if twin else 1):
        x = _q(torch.randn(B, cin, Hin, Win, generator=g), lib.F16)
        w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), lib.F16)
        bias = torch.randn(cout, generator=g)
        xs = torch.zeros(B, Hin, Win, cin + 16, dtype=torch.float16, device=DEV)      # the input is a channel slice of a wider buffer
        xs[..., 8:8 + cin] = _nhwc(x, lib.F16)
        convs.append((xs, w, bias, _act(F.conv2d(x, w, bias, 2, 1), act)))
    assert pack.pack_conv3x3_wreg(convs[0][1], convs[0][2]).numel() == lib.load().maf_conv3s2_wreg_record_bytes(cin, cout)
    res = {}
    for tk, wg in ((7, 0), (7, 1), (7, 3), (72, 3), (1, 0)):
        tk, nbuf = (7, 2) if tk == 72 else (tk, 3)
        outs = [torch.full((B, H, W, 2 * cout), 3.0, dtype=torch.float16, device=DEV) for _ in convs]
        recs = [(pack.pack_conv3x3_wreg(w, b) if tk == 7 else pack.pack_conv3x3(w, 4, lib.F16)).to(DEV) for _, w, b, _ in convs]
        biases = [pack.pack_bias(b, 4).to(DEV) for _, _, b, _ in convs]
        op = _conv_op(lib.OP_CONV3X3S2, lib.F16, B, H, W, cin, cout, act, [(convs[0][0], cin, cin + 16, 8, 0)], outs[0], 2 * cout, cout, recs[0], biases[0],
                      nbuf if tk == 7 else 1, wg if tk == 7 else 4, Hin=Hin, Win=Win)
        op.tile_k = tk
        if twin:
            xs2 = convs[1][0]
            op.aux[0], op.aux[1], op.aux[2], op.aux[3] = xs2.data_ptr(), recs[1].data_ptr(), biases[1].data_ptr(), outs[1].data_ptr()     # base pointers: the slices are shared
        _launch(op)
        for k, (xs, w, b, ref) in enumerate(convs):
            _check(outs[k][..., cout:], ref, lib.F16)
            assert (outs[k][..., :cout] == 3).all(), "wrote outside its slice"
        res[(tk, wg, nbuf)] = [o[..., cout:].float() for o in outs]
    res = {(k_[0], k_[1] if k_[2] == 3 else 72): v for k_, v in res.items()}
    for k in range(len(convs)):
        assert torch.equal(res[(7, 0)][k], res[(7, 1)][k]) and torch.equal(res[(7, 0)][k], res[(7, 3)][k]) and torch.equal(res[(7, 0)][k], res[(7, 72)][k])   # neither the workgroup count nor the buffer count changes a bit
        assert (res[(7, 0)][k] - res[(1, 0)][k]).abs().max() <= 2e-3 * convs[k][3].abs().max() + 2e-3


@pytest.mark.parametrize("cin,cout,act,hw", [(48, 48, lib.ACT_RELU, (36, 50)), (48, 64, lib.ACT_SILU, (64, 64)), (64, 64, lib.ACT_RELU, (22, 34)), (48, 48, lib.ACT_SILU, (160, 160))])
def test_conv3x3_stride2_lds_resident(cin, cout, act, hw):
    """tile_k = 6 (csrc/conv3s2_lds.hip): all weight fragments + the input patch of a 4 x 16 tile in LDS, persistent workgroups; odd input
    sizes, tiles hanging over the map, channel slices on both sides; equals the generic MFMA template to fp32-summation-order noise."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    B, (Hin, Win) = 3, hw
    x = _q(torch.randn(B, cin, Hin, Win, generator=g), lib.F16)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), lib.F16)
    bias = torch.randn(cout, generator=g)
    ref = _act(F.conv2d(x, w, bias, 2, 1), act)
    H, W = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
    xs = torch.zeros(B, Hin, Win, cin + 16, dtype=torch.float16, device=DEV)          # the input is a channel slice of a wider buffer
    xs[..., 8:8 + cin] = _nhwc(x, lib.F16)
    outs = []
    for tk, wg in ((6, 0), (6, 1), (1, 0)):
        out = torch.full((B, H, W, 2 * cout), 3.0, dtype=torch.float16, device=DEV)
        wp = (pack.pack_conv3x3_lds(w, bias) if tk == 6 else pack.pack_conv3x3(w, 4, lib.F16)).to(DEV)
        op = _conv_op(lib.OP_CONV3X3S2, lib.F16, B, H, W, cin, cout, act, [(xs, cin, cin + 16, 8, 0)], out, 2 * cout, cout, wp, pack.pack_bias(bias, 4).to(DEV),
                      4 if tk == 6 else 1, wg if tk == 6 else 4, Hin=Hin, Win=Win)
        op.tile_k = tk
        _launch(op)
        _check(out[..., cout:], ref, lib.F16)
        assert (out[..., :cout] == 3).all(), "wrote outside its slice"
        outs.append(out[..., cout:].float())
    assert torch.equal(outs[0], outs[1])                                             # the workgroup count does not change a bit
    assert (outs[0] - outs[2]).abs().max() <= 2e-3 * ref.abs().max() + 2e-3
    assert wp.numel() * 0 == 0 and pack.pack_conv3x3_lds(w, bias).numel() == lib.load().maf_conv3s2_lds_record_bytes(cin, cout)


@pytest.mark.parametrize("c", [48, 64, 96])
@pytest.mark.parametrize("B,hw", [(3, (36, 52)), (2, (160, 160)), (1, (8, 6)), (2, (70, 34))])
def test_mprep_in_one_launch(B, hw, c):
    """MAF_OP_CONV3X3S2 with nc = c: MPRep = cat(conv1(MaxPool2d(2, 2)(x)), conv2(x)) (common.py:776-792) in ONE launch — tile_k = 6 (csrc/conv3s2_lds.hip, 48 / 64
    channels: the pooled 1x1 + SiLU branch is taken from the patch the 3x3 stride-2 conv stages in LDS) and tile_k = 7 (csrc/conv3s2_wreg.hip, 96 channels: its operand
    is the maximum of four fragments the conv reads anyway).  Against torch in fp32 on the same fp16 operands, against the two separate launches the plan used
    before, tiles hanging over the map, a channel slice as input, nothing written beside the two halves; the conv half is bit-identical to the launch without the branch."""
    g = torch.Generator().manual_seed(B * 100 + hw[0])
    Hin, Win = hw
    tk = 7 if c == 96 else 6
    x = _q(torch.randn(B, c, Hin, Win, generator=g), lib.F16)
    w2 = _q(torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5), lib.F16); b2 = torch.randn(c, generator=g)
    w1 = _q(torch.randn(c, c, 1, 1, generator=g) / c ** 0.5, lib.F16); b1 = torch.randn(c, generator=g)
    ref2 = _act(F.conv2d(x, w2, b2, 2, 1), lib.ACT_RELU)
    ref1 = _act(F.conv2d(F.max_pool2d(x, 2, 2), w1, b1), lib.ACT_SILU)
    H, W = Hin // 2, Win // 2
    xs = torch.zeros(B, Hin, Win, c + 16, dtype=torch.float16, device=DEV)
    xs[..., 8:8 + c] = _nhwc(x, lib.F16)
    rec = (pack.pack_mprep_wreg if tk == 7 else pack.pack_mprep_lds)(w2, b2, w1, b1).to(DEV)
    assert rec.numel() == (lib.load().maf_mprep_wreg_record_bytes if tk == 7 else lib.load().maf_mprep_lds_record_bytes)(c, c, c)
    outs = []
    for pt, wg in (((2, 0), (3, 1), (2, 8)) if tk == 7 else ((4, 0), (4, 1), (4, 3))):     # tile_k = 7: patch buffers, workgroups / 32; tile_k = 6: workgroups / 64
        out = torch.full((B, H, W, 2 * c + 8), 3.0, dtype=torch.float16, device=DEV)
        op = _conv_op(lib.OP_CONV3X3S2, lib.F16, B, H, W, c, c, lib.ACT_RELU, [(xs, c, c + 16, 8, 0)], out, 2 * c + 8, c, rec, pack.pack_bias(b2, 4).to(DEV), pt, wg, Hin=Hin, Win=Win)
        op.tile_k, op.nc, op.reg_stride = tk, c, 0
        _launch(op)
        _check(out[..., c:2 * c], ref2, lib.F16)
        _check(out[..., :c], ref1, lib.F16)
        assert (out[..., 2 * c:] == 3).all(), "wrote outside its pixels' two halves"
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])           # neither the workgroup count nor the buffer count changes a bit
    # the two launches it replaces
    sep = torch.full((B, H, W, 2 * c + 8), 3.0, dtype=torch.float16, device=DEV)
    op2 = _conv_op(lib.OP_CONV3X3S2, lib.F16, B, H, W, c, c, lib.ACT_RELU, [(xs, c, c + 16, 8, 0)], sep, 2 * c + 8, c,
                   (pack.pack_conv3x3_wreg if tk == 7 else pack.pack_conv3x3_lds)(w2, b2).to(DEV), pack.pack_bias(b2, 4).to(DEV), 2 if tk == 7 else 4, 0, Hin=Hin, Win=Win)
    op2.tile_k = tk
    _launch(op2)
    op1 = _conv_op(lib.OP_CONV1X1, lib.F16, B, H, W, c, c, lib.ACT_SILU, [(xs, c, c + 16, 8, lib.SRC_POOL2)], sep, 2 * c + 8, 0, pack.pack_conv1x1(w1, [c], 4, lib.F16).to(DEV), pack.pack_bias(b1, 4).to(DEV), 1, 4)
    _launch(op1)
    assert torch.equal(sep[..., c:2 * c], outs[0][..., c:2 * c])
    assert (sep[..., :c].float() - outs[0][..., :c].float()).abs().max() <= 2e-3 * ref1.abs().max() + 2e-3     # (the bias enters the sum first here, last there)
    # refused: a kernel variant without the branch, the branch's channels not beside / in front of the conv's, an odd input size
    for tk_, hin, off in ((1, Hin, 0), (13 - tk, Hin, 0), (tk, Hin, c), (tk, Hin, 8), (tk, Hin - 1, 0)):
        bad = lib.MafOp.from_buffer_copy(op)
        bad.tile_k, bad.reg_stride, bad.Hin = tk_, off, hin
        if tk_ == 1:
            bad.tile_p, bad.tile_c = 1, 4
        assert lib.load().maf_op_launch(C.byref(bad), torch.cuda.current_stream().cuda_stream) != 0, (tk_, off)


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("k", [3, 5, 7, 9])
@pytest.mark.parametrize("C_,H,W", [(72, 11, 10), (192, 40, 40), (576, 20, 20), (24, 37, 50)])   # ragged tiles, multi-tile, multi-block
def test_dwconv(dt, k, C_, H, W):
    g = torch.Generator().manual_seed(k)
    B = 2
    x = _q(torch.randn(B, C_, H, W, generator=g), dt)
    w = _q(torch.randn(C_, 1, k, k, generator=g) / k, dt)
    bias = torch.randn(C_, generator=g)
    for act in (lib.ACT_SILU, lib.ACT_NONE):
        ref = _act(F.conv2d(x, w, bias, 1, k // 2, 1, C_), act)
        out = torch.zeros(B, H, W, C_, dtype=DT[dt], device=DEV)
        op = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, C_, act, [(_nhwc(x, dt), C_, C_, 0, 0)], out, C_, 0,
                      pack.pack_dw(w, dt).to(DEV), bias.to(DEV), 0, 0)
        op.ksize = k
        _launch(op)
        _check(out, ref, dt)

@pytest.mark.parametrize("k,C_,H,W,th,tw,cb", [(9, 576, 20, 20, 20, 24, 32), (9, 40, 20, 20, 10, 24, 16), (7, 72, 21, 27, 8, 16, 64), (5, 64, 33, 16, 16, 16, 32),
                                               (3, 24, 16, 40, 4, 40, 8), (9, 192, 9, 11, 3, 8, 64), (5, 128, 80, 80, 16, 16, 32), (7, 288, 40, 40, 20, 40, 16)])
@pytest.mark.parametrize("two", [False, True])
def test_dwconv_dot2_variant(k, C_, H, W, th, tw, cb, two):
    """tile_p = -2 (csrc/dwconv_dot2.hip): two taps per instruction (v_dot2_f32_f16) on a pair-interleaved halo tile; odd sizes, tiles hanging over
    the map, odd and even tile heights (one / two strips per lane), slices of wider buffers, two filters per input channel (the head's cls / reg pair);
    against F.conv2d in fp32 on the same fp16 operands and against the v_fma_mix kernel."""
    g = torch.Generator().manual_seed(300 + k + C_)
    B, dt = 2, lib.F16
    cout = 2 * C_ if two else C_
    x = _q(torch.randn(B, C_, H, W, generator=g), dt)
    w = _q(torch.randn(cout, 1, k, k, generator=g) / k, dt)
    bias = torch.randn(cout, generator=g)
    xs = torch.zeros(B, H, W, C_ + 16, dtype=torch.float16, device=DEV)
    xs[..., 8:8 + C_] = _nhwc(x, dt)
    xin = torch.cat([x, x], 1) if two else x
    for act in (lib.ACT_SILU, lib.ACT_NONE):
        ref = _act(F.conv2d(xin, w, bias, 1, k // 2, 1, cout), act)
        outs = []
        for tp in (-2, 0):
            out = torch.full((B, H, W, cout + 8), 3.0, dtype=torch.float16, device=DEV)
            op = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, cout, act, [(xs, C_, C_ + 16, 8, 0)], out, cout + 8, 8,
                          pack.pack_dw(w, dt).to(DEV), bias.to(DEV), tp, tw if tp else 0)
            op.ksize = k
            op.tile_k = th * 256 + cb if tp else 0
            _launch(op)
            _check(out[..., 8:], ref, dt)
            assert (out[..., :8] == 3).all()
            outs.append(out[..., 8:].float())
        assert (outs[0] - outs[1]).abs().max() <= 2e-3 * ref.abs().max() + 2e-3


@pytest.mark.parametrize("k,C_,H,W,th,tw,nw", [(9, 576, 20, 20, 20, 20, 8), (9, 40, 20, 20, 10, 20, 4), (7, 72, 21, 26, 8, 16, 8), (5, 64, 33, 16, 16, 16, 4),
                                               (3, 24, 16, 40, 4, 40, 1), (9, 192, 9, 12, 3, 8, 3), (5, 128, 80, 80, 16, 16, 8), (7, 288, 40, 40, 5, 40, 4),
                                               (5, 128, 80, 80, 4, 80, 8), (3, 16, 6, 10, 8, 12, 2)])
@pytest.mark.parametrize("two", [False, True])
def test_dwconv_pixel_pair_variant(k, C_, H, W, th, tw, nw, two):
    """tile_p = -4 (csrc/dwconv_p2.hip): input stored as pixel pairs (MAF_SRC_PAIRS, pack.pairs_from_nhwc), v_dot2c with the even / odd weight-pair
    sets as scalar operands (pack.pack_dw_pairs), halo planes by DMA; all four kernel sizes (P even and odd: aligned and shifted windows), tiles
    hanging over the map, ragged last pass, workgroups that are not full, a slice of a wider pair buffer, two filters per input channel from one
    gather; against F.conv2d in fp32 on the same fp16 operands and against the v_fma_mix kernel on the NHWC form of the same tensor."""
    g = torch.Generator().manual_seed(500 + k + C_)
    B, dt = 2, lib.F16
    cout = 2 * C_ if two else C_
    x = _q(torch.randn(B, C_, H, W, generator=g), dt)
    w = _q(torch.randn(cout, 1, k, k, generator=g) / k, dt)
    bias = torch.randn(cout, generator=g)
    xs = torch.zeros(B, H, W, C_ + 16, dtype=torch.float16, device=DEV)
    xs[..., 8:8 + C_] = _nhwc(x, dt)
    xp = torch.full((B, H, W // 2, (C_ + 16) * 2), 7.0, dtype=torch.float16, device=DEV)           # pair buffer of stride C_ + 16, slice at channel 8
    xp[..., 16:16 + 2 * C_] = pack.pairs_from_nhwc(_nhwc(x, dt)).reshape(B, H, W // 2, 2 * C_)
    wpairs = pack.pack_dw_pairs(w).to(DEV)
    xin = torch.cat([x, x], 1) if two else x
    for act in (lib.ACT_SILU, lib.ACT_NONE):
        ref = _act(F.conv2d(xin, w, bias, 1, k // 2, 1, cout), act)
        outs = []
        for tp in (-4, 0):
            out = torch.full((B, H, W, cout + 8), 3.0, dtype=torch.float16, device=DEV)
            op = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, cout, act, [(xp if tp else xs, C_, C_ + 16, 8, lib.SRC_PAIRS if tp else 0)], out, cout + 8, 8,
                          pack.pack_dw(w, dt).to(DEV), bias.to(DEV), tp, tw if tp else 0)
            op.ksize = k
            op.tile_k = th * 256 + nw if tp else 0
            if tp:
                op.aux[1] = wpairs.data_ptr()
            _launch(op)
            _check(out[..., 8:], ref, dt)
            assert (out[..., :8] == 3).all()
            outs.append(out[..., 8:].float())
        assert (outs[0] - outs[1]).abs().max() <= 2e-3 * ref.abs().max() + 2e-3
    bad = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, cout, 0, [(xp, C_, C_ + 16, 8, lib.SRC_PAIRS)], out, cout + 8, 8, pack.pack_dw(w, dt).to(DEV), bias.to(DEV), 0, 0)
    bad.ksize = k
    assert lib.load().maf_op_launch(C.byref(bad), torch.cuda.current_stream().cuda_stream) != 0      # a pair source with any other variant is refused


@pytest.mark.parametrize("k,C_,H,W,th,tw,nw", [(9, 288, 20, 20, 10, 20, 4), (9, 192, 20, 20, 10, 20, 2), (7, 128, 40, 40, 10, 20, 4), (5, 128, 80, 80, 16, 16, 4),
                                               (5, 128, 80, 80, 16, 16, 8), (5, 64, 33, 16, 16, 16, 4), (7, 64, 21, 26, 8, 16, 4), (3, 32, 18, 40, 8, 32, 2)])
@pytest.mark.parametrize("two", [False, True])
def test_dwconv_pixel_pair_staged_stores(k, C_, H, W, th, tw, nw, two):
    """tile_k + 128 of the pixel-pair kernel (csrc/dwconv_p2.hip, NF > 0): the workgroup's waves are adjacent channel groups of one tile and the results leave
    through the dead planes as nw x 16-byte runs per pixel.  Same arithmetic as the unstaged form: BIT-identical to it (production tiles of MAF-YOLO-n, tiles
    hanging over the map both ways, two filters per input channel whose results wait in registers), untouched channels beside the slice stay untouched,
    and the fp32 reference bar of the other variants; a tile / wave count the form does not take is refused."""
    g = torch.Generator().manual_seed(900 + k + C_)
    B, dt = 2, lib.F16
    cout = 2 * C_ if two else C_
    x = _q(torch.randn(B, C_, H, W, generator=g), dt)
    w = _q(torch.randn(cout, 1, k, k, generator=g) / k, dt)
    bias = torch.randn(cout, generator=g)
    xp = torch.full((B, H, W // 2, (C_ + 16) * 2), 7.0, dtype=torch.float16, device=DEV)
    xp[..., 16:16 + 2 * C_] = pack.pairs_from_nhwc(_nhwc(x, dt)).reshape(B, H, W // 2, 2 * C_)
    wpairs = pack.pack_dw_pairs(w).to(DEV)
    xin = torch.cat([x, x], 1) if two else x
    for act in (lib.ACT_SILU, lib.ACT_NONE):
        ref = _act(F.conv2d(xin, w, bias, 1, k // 2, 1, cout), act)
        outs = []
        for stg in (128, 0):
            out = torch.full((B, H, W, cout + 16), 3.0, dtype=torch.float16, device=DEV)
            op = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, cout, act, [(xp, C_, C_ + 16, 8, lib.SRC_PAIRS)], out, cout + 16, 8,
                          pack.pack_dw(w, dt).to(DEV), bias.to(DEV), -4, tw)
            op.ksize = k
            op.tile_k = th * 256 + nw + stg
            op.aux[1] = wpairs.data_ptr()
            _launch(op)
            _check(out[..., 8:8 + cout], ref, dt)
            assert (out[..., :8] == 3).all() and (out[..., 8 + cout:] == 3).all()
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
    bad = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, cout, 0, [(xp, C_, C_ + 16, 8, lib.SRC_PAIRS)], out, cout + 16, 8, pack.pack_dw(w, dt).to(DEV), bias.to(DEV), -4, tw)
    bad.ksize = k
    bad.aux[1] = wpairs.data_ptr()
    bad.tile_k = th * 256 + 3 + 128                                    # three waves: not a power of two
    assert lib.load().maf_op_launch(C.byref(bad), torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize("k,C_,H,W", [(7, 72, 21, 27), (9, 40, 20, 20), (5, 64, 33, 16), (3, 24, 16, 40), (9, 192, 9, 11)])
def test_dwconv_matrix_core_variant(k, C_, H, W):
    """tile_p = -1: depth-wise conv as block-diagonal Toeplitz MFMAs (csrc/dwconv_mfma.hip); slices of wider buffers."""
    g = torch.Generator().manual_seed(100 + k)
    B, dt = 2, lib.F16
    x = _q(torch.randn(B, C_, H, W, generator=g), dt)
    w = _q(torch.randn(C_, 1, k, k, generator=g) / k, dt)
    bias = torch.randn(C_, generator=g)
    toe = pack.pack_dw_toeplitz(w).to(DEV)
    xs = torch.zeros(B, H, W, C_ + 16, dtype=torch.float16, device=DEV)
    xs[..., 8:8 + C_] = _nhwc(x, dt)
    for act in (lib.ACT_SILU, lib.ACT_NONE):
        ref = _act(F.conv2d(x, w, bias, 1, k // 2, 1, C_), act)
        out = torch.full((B, H, W, C_ + 8), 3.0, dtype=torch.float16, device=DEV)
        op = _conv_op(lib.OP_DWCONV, dt, B, H, W, C_, C_, act, [(xs, C_, C_ + 16, 8, 0)], out, C_ + 8, 8,
                      pack.pack_dw(w, dt).to(DEV), bias.to(DEV), -1, 0)
        op.ksize = k
        op.aux[0] = toe.data_ptr()
        _launch(op)
        _check(out[..., 8:], ref, dt)
        assert (out[..., :8] == 3).all()



@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("in_dt", [lib.F32, lib.F16, lib.U8])
def test_stem(dt, in_dt):
    g = torch.Generator().manual_seed(3)
    B, Hin, Win, cout = 2, 32, 64, 24
    if in_dt == lib.U8:
        img = torch.randint(0, 256, (B, 3, Hin, Win), generator=g, dtype=torch.uint8)
        x = img.float() / 255
        dimg = img.to(DEV)
    else:
        x = _q(torch.rand(B, 3, Hin, Win, generator=g), in_dt)
        dimg = x.to(DT[in_dt]).to(DEV)
    w = torch.randn(cout, 3, 3, 3, generator=g) / 5.0
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, 2, 1))
    out = torch.zeros(B, Hin // 2, Win // 2, cout, dtype=DT[dt], device=DEV)
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = lib.OP_STEM, dt, in_dt, lib.ACT_RELU
    op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout = B, Hin // 2, Win // 2, Hin, Win, 3, cout
    op.nsrc = 1
    op.src[0].ptr = dimg.data_ptr()
    op.out, op.out_stride, op.out_coff = out.data_ptr(), cout, 0
    ws, bs = pack.pack_stem(w).to(DEV), bias.to(DEV)
    op.w, op.bias = ws.data_ptr(), bs.data_ptr()
    _launch(op)
    _check(out, ref, dt)


@pytest.mark.parametrize("dt", [lib.F32, lib.F16])
@pytest.mark.parametrize("H,W", [(10, 7), (20, 20), (66, 70)])       # LDS kernel / brute-force fallback (> 64x64)
def test_sppf_pool(dt, H, W):
    g = torch.Generator().manual_seed(4)
    B, c_ = 2, 48
    x = _q(torch.randn(B, c_, H, W, generator=g), dt)
    y1 = F.max_pool2d(x, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    buf = torch.zeros(B, H, W, 4 * c_, dtype=DT[dt], device=DEV)
    buf[..., :c_] = _nhwc(x, dt)
    op = lib.MafOp()
    op.kind, op.dtype, op.B, op.H, op.W, op.nsrc = lib.OP_SPPF_POOL, dt, B, H, W, 1
    op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff = buf.data_ptr(), c_, 4 * c_, 0
    op.out, op.out_stride, op.out_coff = buf.data_ptr(), 4 * c_, c_
    _launch(op)
    got = buf.float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, torch.cat([x, y1, y2, y3], 1))          # max pooling is exact in any precision


def test_decode():
    from oracle import maf_oracle as O
    g = torch.Generator().manual_seed(6)
    B, nc = 2, 80
    dims = [(12, 20), (6, 10), (3, 5)]
    heads, cls_d, reg_d = [], [], []
    for h, w in dims:
        cls = torch.rand(B, nc, h, w, generator=g)
        reg = torch.randn(B, 68, h, w, generator=g) * 2
        heads.append((torch.zeros(B, 1, h, w), cls, reg))
        cls_d.append(cls.permute(0, 2, 3, 1).contiguous().to(DEV)); reg_d.append(reg.permute(0, 2, 3, 1).contiguous().to(DEV))
    ref = O.decode(heads)
    A = sum(h * w for h, w in dims)
    out = torch.zeros(B, A, 5 + nc, device=DEV)
    op = lib.MafOp()
    op.kind, op.B, op.nsrc = lib.OP_DECODE, B, 3
    for l, (h, w) in enumerate(dims):
        op.src[l].ptr, op.reg[l] = cls_d[l].data_ptr(), reg_d[l].data_ptr()
        op.lvl_h[l], op.lvl_w[l], op.lvl_stride[l] = h, w, float(O.STRIDES[l])
    op.reg_stride, op.nc, op.reg_max = 68, nc, 16
    op.out = out.data_ptr()
    _launch(op)
    got = out.cpu()
    np.testing.assert_allclose(got[..., :4].numpy(), ref[..., :4].numpy(), rtol=1e-5, atol=1e-3)
    assert torch.equal(got[..., 4:], ref[..., 4:])


def test_bad_arguments_are_rejected():
    op = lib.MafOp()
    op.kind, op.dtype, op.B, op.H, op.W, op.Cin, op.Cout, op.nsrc = lib.OP_CONV1X1, lib.F16, 1, 4, 4, 12, 16, 1
    x = torch.zeros(64, device=DEV)
    op.src[0].ptr, op.src[0].C, op.src[0].stride = x.data_ptr(), 12, 12          # 12 is not a multiple of 8
    op.out, op.w, op.bias, op.out_stride, op.tile_p, op.tile_c = x.data_ptr(), x.data_ptr(), x.data_ptr(), 16, 1, 2
    assert lib.load().maf_op_launch(C.byref(op), None) == -1
    assert b"multiples" in lib.load().maf_last_error()


@pytest.mark.parametrize("c,k,H,W", [(64, 5, 21, 27), (24, 3, 21, 27), (48, 5, 32, 16), (64, 7, 21, 27), (32, 9, 20, 20), (16, 3, 40, 33), (40, 7, 16, 16)])
def test_fused_bottleneck(c, k, H, W):
    """MAF_OP_BOTTLENECK == Conv1x1+SiLU -> DW k x k + SiLU -> Conv1x1+SiLU (DepthBottleneckUni deploy form, common.py:918-927)."""
    g = torch.Generator().manual_seed(c + k)
    B = 2                                                 # 21 x 27: ragged 16 x 16 tiles on both axes
    mid = 3 * c
    x = torch.randn(B, c, H, W, generator=g).half().float()
    w1 = (torch.randn(mid, c, 1, 1, generator=g) / c ** 0.5).half().float(); b1 = torch.randn(mid, generator=g) * 0.3
    wd = (torch.randn(mid, 1, k, k, generator=g) / k).half().float(); bd = torch.randn(mid, generator=g) * 0.3
    w2 = (torch.randn(c, mid, 1, 1, generator=g) / mid ** 0.5).half().float(); b2 = torch.randn(c, generator=g) * 0.3
    t1 = F.silu(F.conv2d(x, w1, b1)).half().float()                     # the separate kernels also store T1/T2 in fp16
    t2 = F.silu(F.conv2d(t1, wd, bd, 1, k // 2, 1, mid)).half().float()
    ref = F.silu(F.conv2d(t2, w2, b2))
    rec, b2p, nmb, ct2 = pack.pack_bottleneck(w1, b1, wd, bd, w2, b2)
    assert rec.shape[1] == lib.load().maf_bottleneck_record_bytes(k, c, c)
    dev = [rec.to(DEV), b2p.to(DEV)]
    stride = 2 * c + 16
    buf = torch.zeros(B, H, W, stride, dtype=torch.float16, device=DEV)
    buf[..., 8:8 + c] = _nhwc(x, lib.F16)
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = lib.OP_BOTTLENECK, lib.F16, lib.F16, lib.ACT_SILU
    op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc = B, H, W, c, c, k, 1
    op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff = buf.data_ptr(), c, stride, 8
    op.out, op.out_stride, op.out_coff = buf.data_ptr(), stride, 8 + c
    op.tile_p, op.tile_c, op.tile_k = 16, 16, nmb
    op.w, op.bias = dev[0].data_ptr(), dev[1].data_ptr()
    _launch(op)
    _check(buf[..., 8 + c:8 + 2 * c], ref, lib.F16)
    assert (buf[..., :8] == 0).all() and (buf[..., 8 + 2 * c:] == 0).all()
    assert torch.equal(buf[..., 8:8 + c].float().cpu().permute(0, 3, 1, 2), x)         # input slice untouched


@pytest.mark.parametrize("c,k,c3,nsrc,H,W", [(24, 3, 48, 2, 21, 27), (48, 5, 96, 2, 32, 16), (64, 5, 128, 2, 21, 27), (64, 5, 128, 2, 80, 80),
                                             (32, 3, 64, 3, 21, 27), (64, 5, 128, 3, 19, 33), (48, 3, 96, 3, 16, 40)])
def test_fused_bottleneck_with_closing_conv(c, k, c3, nsrc, H, W):
    """MAF_OP_BOTTLENECK with nc = C3: the last DepthBottleneckUni of a RepHDW block and the block's closing conv2(cat(x1, x2, .., y)) + SiLU
    (common.py:918-927, 938-946) in one launch == the two steps on their own (y rounded to fp16 where the separate kernels store it)."""
    g = torch.Generator().manual_seed(7 * c + k + c3)
    B, mid = 2, 3 * c
    slots = [torch.randn(B, c, H, W, generator=g).half().float() for _ in range(nsrc)]          # concat slots in front of y; the last one feeds the bottleneck
    x = slots[-1]
    w1 = (torch.randn(mid, c, 1, 1, generator=g) / c ** 0.5).half().float(); b1 = torch.randn(mid, generator=g) * 0.3
    wd = (torch.randn(mid, 1, k, k, generator=g) / k).half().float(); bd = torch.randn(mid, generator=g) * 0.3
    w2 = (torch.randn(c, mid, 1, 1, generator=g) / mid ** 0.5).half().float(); b2 = torch.randn(c, generator=g) * 0.3
    w3 = (torch.randn(c3, (nsrc + 1) * c, 1, 1, generator=g) / ((nsrc + 1) * c) ** 0.5).half().float(); b3 = torch.randn(c3, generator=g) * 0.3
    t1 = F.silu(F.conv2d(x, w1, b1)).half().float()
    t2 = F.silu(F.conv2d(t1, wd, bd, 1, k // 2, 1, mid)).half().float()
    y = F.silu(F.conv2d(t2, w2, b2)).half().float()
    ref = F.silu(F.conv2d(torch.cat(slots + [y], 1), w3, b3))
    L = lib.load()
    assert L.maf_bottleneck_tail_supported(k, c, nsrc, c3) == 1
    rec, b2p, nmb, ct2 = pack.pack_bottleneck(w1, b1, wd, bd, w2, b2)
    rec3 = pack.pack_bottleneck_tail(w3, b3, c, nsrc)
    assert rec3.numel() == L.maf_bottleneck_tail_record_bytes(c, nsrc, c3)
    dev = [rec.to(DEV), b2p.to(DEV), rec3.to(DEV)]
    stride = nsrc * c + 8                                               # the slots interleaved in one buffer behind 8 guard channels, as the engine lays them out
    buf = torch.zeros(B, H, W, stride, dtype=torch.float16, device=DEV)
    for i, t in enumerate(slots):
        buf[..., 8 + i * c:8 + (i + 1) * c] = _nhwc(t, lib.F16)
    keep = buf.clone()
    out = torch.full((B, H, W, c3 + 8), 3.0, dtype=torch.float16, device=DEV)
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = lib.OP_BOTTLENECK, lib.F16, lib.F16, lib.ACT_SILU
    op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc, op.nc = B, H, W, c, c, k, nsrc, c3
    order = [nsrc - 1] + list(range(nsrc - 1))                          # src[0] = the bottleneck's input, src[1..] = the slots in front of it
    for j, i in enumerate(order):
        op.src[j].ptr, op.src[j].C, op.src[j].stride, op.src[j].coff, op.src[j].mode = buf.data_ptr(), c, stride, 8 + i * c, lib.SRC_DIRECT
    op.out, op.out_stride, op.out_coff = out.data_ptr(), c3 + 8, 8
    op.tile_p, op.tile_c, op.tile_k = 16, 16, nmb
    op.w, op.bias = dev[0].data_ptr(), dev[1].data_ptr()
    op.aux[0] = dev[2].data_ptr()
    _launch(op)
    _check(out[..., 8:], ref, lib.F16)
    assert (out[..., :8] == 3).all() and torch.equal(buf, keep)         # nothing but the closing conv's channels is written
    op.nc = 32                                                          # no such instantiation: a clean error, no launch
    assert L.maf_op_launch(C.byref(op), None) < 0


@pytest.mark.parametrize("c,k,H,W", [(96, 7, 21, 27), (192, 9, 20, 20), (24, 3, 33, 18), (128, 5, 16, 32), (72, 5, 40, 40)])
def test_conv1_dw_partial_fusion(c, k, H, W):
    """MAF_OP_CONV1DW == Conv1x1+SiLU -> DW k x k + SiLU (first half of DepthBottleneckUni, common.py:905-909) for any width."""
    g = torch.Generator().manual_seed(3 * c + k)
    B, mid = 2, 3 * c
    x = torch.randn(B, c, H, W, generator=g).half().float()
    w1 = (torch.randn(mid, c, 1, 1, generator=g) / c ** 0.5).half().float(); b1 = torch.randn(mid, generator=g) * 0.3
    wd = (torch.randn(mid, 1, k, k, generator=g) / k).half().float(); bd = torch.randn(mid, generator=g) * 0.3
    t1 = F.silu(F.conv2d(x, w1, b1)).half().float()
    ref = F.silu(F.conv2d(t1, wd, bd, 1, k // 2, 1, mid))
    rec, nmb = pack.pack_conv1dw(w1, b1, wd, bd)
    assert rec.shape[1] == lib.load().maf_conv1dw_record_bytes(k, c) and nmb == -(-mid // 32)
    recd = rec.to(DEV)
    xs = torch.zeros(B, H, W, c + 8, dtype=torch.float16, device=DEV)
    xs[..., 8:] = _nhwc(x, lib.F16)
    out = torch.full((B, H, W, mid + 8), 2.0, dtype=torch.float16, device=DEV)
    op = lib.MafOp()
    op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV1DW, lib.F16, lib.F16, lib.ACT_SILU
    op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc = B, H, W, c, mid, k, 1
    op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff = xs.data_ptr(), c, c + 8, 8
    op.out, op.out_stride, op.out_coff = out.data_ptr(), mid + 8, 8
    op.w = recd.data_ptr()
    _launch(op)
    _check(out[..., 8:], ref, lib.F16)
    assert (out[..., :8] == 2).all()


@pytest.mark.parametrize("dt", [lib.F16, lib.F32])
@pytest.mark.parametrize("tk,pt,ct", [(1, 2, 4), (2, 1, 4), (4, 1, 4), (1, 1, 8), (8, 2, 4), (8, 1, 8)])
def test_conv3x3s2_twin_launch(dt, tk, pt, ct):
    """MAF_OP_CONV3X3S2 with aux = {src, w, bias, out} of a second conv: both results equal their own fp32 references."""
    if dt == lib.F32 and tk in (2, 8):
        pytest.skip("tile_k = 2 is an fp16 variant")
    g = torch.Generator().manual_seed(tk * 10 + pt)
    B, Hin, Win, cin, cout = 2, 22, 18, 64, 64
    H, W = Hin // 2, Win // 2
    xs = [_q(torch.randn(B, cin, Hin, Win, generator=g), dt) for _ in range(2)]
    ws = [_q(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, dt) for _ in range(2)]
    bs = [torch.randn(cout, generator=g) * 0.2 for _ in range(2)]
    refs = [F.silu(F.conv2d(xs[i], ws[i], bs[i], 2, 1)) for i in range(2)]
    xd = [_nhwc(x, dt) for x in xs]
    wd = [pack.pack_conv3x3(w, ct, dt).to(DEV) for w in ws]
    bd = [pack.pack_bias(b, ct).to(DEV) for b in bs]
    outs = [torch.zeros(B, H, W, cout, dtype=DT[dt], device=DEV) for _ in range(2)]
    op = _conv_op(lib.OP_CONV3X3S2, dt, B, H, W, cin, cout, lib.ACT_SILU, [(xd[0], cin, cin, 0, lib.SRC_DIRECT)], outs[0], cout, 0, wd[0], bd[0], pt, ct, Hin=Hin, Win=Win)
    op.tile_k = tk
    op.aux[0], op.aux[1], op.aux[2], op.aux[3] = xd[1].data_ptr(), wd[1].data_ptr(), bd[1].data_ptr(), outs[1].data_ptr()
    _launch(op)
    for i in range(2):
        _check(outs[i], refs[i], dt)
    op.aux[1] = None                                          # a half-specified twin is rejected
    assert lib.load().maf_op_launch(C.byref(op), None) == -1
