"""`torch.ops.mafyolo.*` — the PyTorch-ROCm custom-op layer of the hot path (BASELINE.json north_star).

    from maf_yolo_amd import torch_ops
    ops = torch_ops.load()                                        # torch.ops.load_library("maf-yolo_amd/libmafyolo_torch.so") + registrations
    y = ops.conv1x1_bias_act(x, w, b, torch_ops.ACT_SILU)          # Conv.forward_fuse (yolov6/layers/common.py:49-50)
    y = ops.conv3x3s2_bias_act(x, w, b, torch_ops.ACT_RELU)        # RepVGGBlock deploy forward (:216-217) / ConvWrapper (:76-83)
    y = ops.dwconv_bias_act(x, w, b, torch_ops.ACT_NONE)           # merged UniRepLKNetBlock (:3085-3100)
    pred = ops.head_decode(cls, reg, [8., 16., 32.])               # Detect_yaml eval branch (yolov6/models/yolo.py:355-396)
    rows, counts = ops.decode_nms(pred, 0.03, 0.65, False, True, 300, None)     # non_max_suppression (yolov6/utils/nms.py:31-105)
    dets = torch_ops.non_max_suppression(pred, 0.03, 0.65, multi_label=True)    # ... as the reference's list of [n_i, 6] tensors

The C++ side (csrc/torch_ops.cpp) defines the schemas and the HIP ("CUDA" dispatch key) implementations, which marshal at::Tensor into
the C-ABI of libmafyolo_hip.so on the current stream.  Registered here, through torch.library:
  * autograd for the three convolutions with act = NONE (the train-form graph keeps conv, BatchNorm and activation apart:
    data gradient and weight gradient are ops of the same library — conv1x1_dgrad, conv3x3s2_dgrad, conv_wgrad, dwconv_dgrad, dwconv_wgrad);
  * the autocast rule of a convolution (inputs cast to fp16 under torch.autocast("cuda"));
  * fake (meta) kernels, so the ops trace under torch.compile / FakeTensor.
The engine (engine.py) and the training layers (train_ops.py) bind the same C-ABI directly through ctypes: one launch list per forward
needs no dispatcher round trip per op; this module is the surface a PyTorch program — the reference's evaler / trainer — calls op by op.
"""
import os

import torch

from . import lib

ACT_NONE, ACT_RELU, ACT_SILU, ACT_SIGMOID = lib.ACT_NONE, lib.ACT_RELU, lib.ACT_SILU, lib.ACT_SIGMOID
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmafyolo_torch.so")
OPS = ("conv1x1_bias_act", "conv3x3s2_bias_act", "dwconv_bias_act", "conv1x1_dgrad", "conv3x3s2_dgrad", "conv_wgrad", "dwconv_dgrad", "dwconv_wgrad",
       "head_decode", "decode_nms", "mprep", "sppf", "bn_act", "bn_act_backward")
_registered = False


def _out_hw(h, w):
    return (h - 1) // 2 + 1, (w - 1) // 2 + 1


def _cl(like, shape, dtype=None):
    """Empty channels_last tensor of `shape` with `like`'s device (fake kernels)."""
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device, memory_format=torch.channels_last)


def _register():
    L = torch.library

    # ---- fake kernels: shapes / dtypes only
    @L.register_fake("mafyolo::conv1x1_bias_act")
    def _(x, w, bias, act):
        return _cl(x, (x.shape[0], w.shape[0], x.shape[2], x.shape[3]))

    @L.register_fake("mafyolo::conv3x3s2_bias_act")
    def _(x, w, bias, act):
        return _cl(x, (x.shape[0], w.shape[0], *_out_hw(x.shape[2], x.shape[3])))

    @L.register_fake("mafyolo::dwconv_bias_act")
    def _(x, w, bias, act):
        return torch.empty_like(x, memory_format=torch.channels_last)

    @L.register_fake("mafyolo::conv1x1_dgrad")
    def _(dy, w):
        return _cl(dy, (dy.shape[0], w.shape[1], dy.shape[2], dy.shape[3]))

    @L.register_fake("mafyolo::conv3x3s2_dgrad")
    def _(dy, w, H, W):
        return _cl(dy, (dy.shape[0], w.shape[1], H, W))

    @L.register_fake("mafyolo::conv_wgrad")
    def _(x, dy, ksize, stride):
        return x.new_empty((dy.shape[1], x.shape[1], ksize, ksize), dtype=torch.float32)

    @L.register_fake("mafyolo::dwconv_dgrad")
    def _(dy, w):
        return torch.empty_like(dy, memory_format=torch.channels_last)

    @L.register_fake("mafyolo::dwconv_wgrad")
    def _(x, dy, k):
        return x.new_empty((x.shape[1], 1, k, k), dtype=torch.float32)

    @L.register_fake("mafyolo::head_decode")
    def _(cls, reg, strides):
        A = sum(c.shape[2] * c.shape[3] for c in cls)
        return cls[0].new_empty((cls[0].shape[0], A, 5 + cls[0].shape[1]), dtype=torch.float32)

    @L.register_fake("mafyolo::decode_nms")
    def _(pred, conf_thres, iou_thres, agnostic, multi_label, max_det, classes):
        return pred.new_empty((pred.shape[0], max_det, 6), dtype=torch.float32), pred.new_empty((pred.shape[0],), dtype=torch.int32)

    @L.register_fake("mafyolo::mprep")
    def _(x, w1, b1, w3, b3):
        return _cl(x, (x.shape[0], w1.shape[0] + w3.shape[0], x.shape[2] // 2, x.shape[3] // 2))

    @L.register_fake("mafyolo::sppf")
    def _(x, w1, b1, w2, b2):
        return _cl(x, (x.shape[0], w2.shape[0], x.shape[2], x.shape[3]))

    @L.register_fake("mafyolo::bn_act")
    def _(x, gamma, beta, running_mean, running_var, eps, momentum, act):
        c = x.shape[1]
        st = lambda n: x.new_empty((n,), dtype=torch.float32)
        return (torch.empty_like(x, memory_format=torch.channels_last), st(c), st(c), st(c if running_mean is not None else 0), st(c if running_var is not None else 0))

    @L.register_fake("mafyolo::bn_act_backward")
    def _(x, dz, gamma, beta, save_mean, save_rstd, act):
        c = x.shape[1]
        return torch.empty_like(x, memory_format=torch.channels_last), x.new_empty((c,), dtype=torch.float32), x.new_empty((c,), dtype=torch.float32)

    # ---- BatchNorm(train) + activation: backward = the bn_act_backward op (recomputes the pre-activation from x and the saved statistics)
    def _bn_setup(ctx, inputs, output):
        x, gamma, beta, rm, rv, eps, momentum, act = inputs
        ctx.save_for_backward(x, gamma, beta, output[1], output[2])
        ctx.act = act

    def _bn_bwd(ctx, dy, dmean, drstd, drm, drv):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dg, db = torch.ops.mafyolo.bn_act_backward(x, dy, gamma, beta, mean, rstd, ctx.act)
        return dx, dg.to(gamma.dtype), db.to(beta.dtype), None, None, None, None, None

    L.register_autograd("mafyolo::bn_act", _bn_bwd, setup_context=_bn_setup)

    # ---- autograd (act = NONE only: conv, BatchNorm and activation are separate layers of the train-form graph)
    ops = torch.ops.mafyolo

    def _setup(ctx, inputs, output):
        x, w, bias, act = inputs
        if act != ACT_NONE and (x.requires_grad or w.requires_grad):
            raise RuntimeError("mafyolo convolution ops are differentiable with act = NONE (apply BatchNorm / activation as their own layers)")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None

    def _wgrad(x, dy, w, ksize, stride):
        if x.dtype == torch.float16:
            return ops.conv_wgrad(x, dy.to(x.dtype), ksize, stride).to(w.dtype)
        if ksize == 1 and stride == 1:                       # fp32 parity mode: the framework's GEMM
            return torch.mm(dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1]).t(), x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])).reshape(w.shape).to(w.dtype)
        return torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=stride, padding=ksize // 2).to(w.dtype)

    def _bwd_1x1(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dx = ops.conv1x1_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = _wgrad(x, dy, w, 1, 1) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3), dtype=torch.float32) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None

    def _bwd_3x3(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dx = ops.conv3x3s2_dgrad(dy, w, x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = _wgrad(x, dy, w, 3, 2) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3), dtype=torch.float32) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None

    def _bwd_dw(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dx = ops.dwconv_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = ops.dwconv_wgrad(x, dy, w.shape[-1]).to(w.dtype) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3), dtype=torch.float32) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None

    L.register_autograd("mafyolo::conv1x1_bias_act", _bwd_1x1, setup_context=_setup)
    L.register_autograd("mafyolo::conv3x3s2_bias_act", _bwd_3x3, setup_context=_setup)
    L.register_autograd("mafyolo::dwconv_bias_act", _bwd_dw, setup_context=_setup)

    # ---- autocast: convolutions run in the lower precision (the kernels are fp16 / fp32: fp16 is the autocast type of this path)
    for name in ("conv1x1_bias_act", "conv3x3s2_bias_act", "dwconv_bias_act"):
        L.register_autocast("mafyolo::" + name, "cuda", torch.float16)


def load():
    """torch.ops.load_library + the registrations above (once).  Raises MafError if either library has not been built."""
    global _registered
    lib.load()                                               # the C-ABI library the op layer links against (no CPU fallback: raises if absent)
    if not os.path.exists(LIB_PATH):
        raise lib.MafError("libmafyolo_torch.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    if not _registered:
        torch.ops.load_library(LIB_PATH)
        _register()
        _registered = True
    return torch.ops.mafyolo


def bn_act_(x, bn, act=ACT_NONE):
    """act(bn(x)) for an nn.BatchNorm2d in training mode through torch.ops.mafyolo.bn_act (functional), with the module's running statistics and
    batch counter updated like nn.BatchNorm2d does (Conv.forward, yolov6/layers/common.py:44-47)."""
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    y, _, _, nrm, nrv = load().bn_act(x, bn.weight, bn.bias, rm, rv, float(bn.eps), float(bn.momentum), int(act))
    if rm is not None:
        with torch.no_grad():
            rm.copy_(nrm); rv.copy_(nrv); bn.num_batches_tracked.add_(1)
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """The reference's post-processing call (yolov6/utils/nms.py:31) over torch.ops.mafyolo.decode_nms: same arguments, same list of
    [n_i, 6] tensors; thresholds outside [0, 1] raise (AssertionError, like nms.py:50-51)."""
    assert 0 <= conf_thres <= 1, f'conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided.'
    assert 0 <= iou_thres <= 1, f'iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided.'
    rows, counts = load().decode_nms(prediction, float(conf_thres), float(iou_thres), bool(agnostic), bool(multi_label), int(max_det),
                                     None if classes is None else [int(c) for c in classes])
    return [rows[b, :n] for b, n in enumerate(counts.tolist())]
