"""Eval forward of the deploy-form graph as a SEQUENCE OF `torch.ops.mafyolo.*` CALLS (the hot path as PyTorch-ROCm custom ops).

`Model(..., dispatch="ops")` / `model.dispatch = "ops"` routes `Model.forward` (eval) through here instead of the one-call HIP engine: every node of
the YAML graph (yolov6/models/yolo.py:186-201) becomes one or a few dispatcher-visible ops on channels_last tensors —

    RepVGGBlock  -> conv3x3s2_bias_act(ReLU)        Conv / ConvWrapper -> conv1x1_bias_act / conv3x3s2_bias_act (SiLU)
    MPRep        -> mprep                           SPPF               -> sppf
    RepHDW       -> conv1x1 -> [conv1x1 -> dwconv(SiLU) -> conv1x1] x depth -> cat -> conv1x1        (DepthBottleneckUni, common.py:898-946)
    Head_DepthUni-> conv1x1, dwconv, conv1x1, conv1x1 (sigmoid / none)                              Detect eval branch -> head_decode

with torch.cat / nearest upsampling as the framework's own ops, so `torch.compile(fullgraph=True)` / export trace the model through the fake
kernels registered in torch_ops.py.  Same kernels, same arithmetic as the engine's UNFUSED plan; it is the op-by-op surface (a dispatcher round
trip, an output allocation and a weight packing per op), not the fast path — the engine stays the default.
Weights: the deploy algebra (`layers.*.fused()`) evaluated once per weight version and cached."""
import torch
import torch.nn.functional as F

from . import torch_ops
from .torch_ops import ACT_NONE, ACT_RELU, ACT_SILU, ACT_SIGMOID


def deploy_weights(model, dtype):
    """{node index: tuple of (w, b) pairs in the order the node's ops take them} — detached, in `dtype` (fp32 biases)."""
    out = {}
    with torch.no_grad():
        def wb(t):
            w, b = t
            return w.detach().to(dtype).contiguous(), b.detach().float().contiguous()
        for nd, m in zip(model.nodes, model.backbone):
            k = nd.kind
            if k == "repvgg":
                out[nd.i] = (wb(m.fused()),)
            elif k == "rephdw":
                ws = [wb(m.conv1.fused())]
                for blk in m.m:
                    ws += [wb(blk.conv1.fused()), wb(blk.conv2.fused()), wb(blk.one_conv.fused())]
                ws.append(wb(m.conv2.fused()))
                out[nd.i] = tuple(ws)
            elif k == "mprep":
                out[nd.i] = (wb(m.conv1.fused()), wb(m.conv2.fused()))
            elif k == "sppf":
                out[nd.i] = (wb(m.cv1.fused()), wb(m.cv2.fused()))
            elif k == "cw":
                out[nd.i] = (wb(m.block.fused()),)
            elif k == "head":
                out[nd.i] = (wb(m.stem.fused()), wb(m.cls_conv.fused()), wb(m.cls_conv_s.fused()), wb((m.cls_pred.weight, m.cls_pred.bias)),
                             wb(m.reg_conv.fused()), wb(m.reg_conv_s.fused()), wb((m.reg_pred.weight, m.reg_pred.bias)))
    return out


def forward(model, x, weights, ops=None, strides=None):
    """x [B, 3, H, W] fp16 / fp32 on the HIP device -> (pred fp32 [B, A, 5 + nc], [(stem, cls, reg)] x 3).
    `ops` (torch.ops.mafyolo after torch_ops.load()) and `strides` (floats) are resolved here when not given; a caller that traces this function
    (Model.traceable) resolves them OUTSIDE the trace — library loading and tensor.tolist() are not traceable."""
    if ops is None:
        ops = torch_ops.load()
    if strides is None:
        strides = [float(s) for s in model.detect.stride.tolist()]
    y, heads = [], []
    for nd, m in zip(model.nodes, model.backbone):
        src = [y[j] for j in nd.sources()] if nd.i > 0 else [x]
        t = src[0]
        k = nd.kind
        w = weights.get(nd.i)
        if k == "repvgg":
            o = ops.conv3x3s2_bias_act(t, w[0][0], w[0][1], ACT_RELU)
        elif k == "rephdw":
            c_ = m.c_
            z = ops.conv1x1_bias_act(t, w[0][0], w[0][1], ACT_SILU)
            outs = [z[:, :c_], z[:, c_:]]
            for d in range(len(m.m)):
                (w1, b1), (wd, bd), (w2, b2) = w[1 + 3 * d: 4 + 3 * d]
                u = ops.conv1x1_bias_act(outs[-1], w1, b1, ACT_SILU)
                u = ops.dwconv_bias_act(u, wd, bd, ACT_SILU)
                outs.append(ops.conv1x1_bias_act(u, w2, b2, ACT_SILU))
            o = ops.conv1x1_bias_act(torch.cat(outs, 1), w[-1][0], w[-1][1], ACT_SILU)
        elif k == "mprep":
            o = ops.mprep(t, w[0][0], w[0][1], w[1][0], w[1][1])
        elif k == "sppf":
            o = ops.sppf(t, w[0][0], w[0][1], w[1][0], w[1][1])
        elif k == "cw":
            o = ops.conv3x3s2_bias_act(t, w[0][0], w[0][1], ACT_SILU)
        elif k == "concat":
            o = torch.cat(src, 1)
        elif k == "up":
            o = F.interpolate(t, scale_factor=2.0, mode="nearest")
        elif k == "head":
            s_ = ops.conv1x1_bias_act(t, w[0][0], w[0][1], ACT_SILU)
            c = ops.conv1x1_bias_act(ops.dwconv_bias_act(s_, w[1][0], w[1][1], ACT_NONE), w[2][0], w[2][1], ACT_SILU)
            cls = ops.conv1x1_bias_act(c, w[3][0], w[3][1], ACT_SIGMOID)
            r = ops.conv1x1_bias_act(ops.dwconv_bias_act(s_, w[4][0], w[4][1], ACT_NONE), w[5][0], w[5][1], ACT_SILU)
            reg = ops.conv1x1_bias_act(r, w[6][0], w[6][1], ACT_NONE)
            heads.append((s_, cls, reg))
            o = None
        elif k == "out":
            o = None
        else:
            raise NotImplementedError(k)
        y.append(o)
    pred = ops.head_decode([h[1] for h in heads], [h[2] for h in heads], list(strides))
    return pred, heads
