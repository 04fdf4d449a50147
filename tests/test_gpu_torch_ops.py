"""-m gpu: torch.ops.mafyolo.* (csrc/torch_ops.cpp + maf_yolo_amd/torch_ops.py) — the PyTorch-ROCm custom-op surface of SURVEY.md 8(b) —
against plain PyTorch fp32 references: forward with the fused epilogues, autograd (data + weight + bias gradients), the autocast rule,
decode and NMS against the oracle / the ctypes path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import maf_yolo_amd as M
from maf_yolo_amd import torch_ops
from oracle import maf_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ACTS = {torch_ops.ACT_NONE: lambda y: y, torch_ops.ACT_RELU: F.relu, torch_ops.ACT_SILU: F.silu, torch_ops.ACT_SIGMOID: torch.sigmoid}


def _rel(a, b):
    return (a.float().cpu() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ops():
    return torch_ops.load()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 4e-3)])
@pytest.mark.parametrize("act", sorted(ACTS))
def test_forward_ops_with_fused_epilogue(ops, dtype, tol, act):
    g = torch.Generator().manual_seed(act)
    x = torch.randn(2, 64, 11, 14, generator=g).to(dtype)
    w1, b1 = torch.randn(81, 64, 1, 1, generator=g) / 8, torch.randn(81, generator=g)            # an odd class count: padded inside
    w3, b3 = torch.randn(48, 64, 3, 3, generator=g) / 24, torch.randn(48, generator=g)
    wd, bd = torch.randn(64, 1, 7, 7, generator=g) / 7, torch.randn(64, generator=g)
    q = (lambda t: t.half().float()) if dtype == torch.float16 else (lambda t: t)
    xr = x.float()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    got = ops.conv1x1_bias_act(xd, w1.to(DEV), b1.to(DEV), act)
    assert got.dtype == dtype and got.shape == (2, 81, 11, 14) and _rel(got, ACTS[act](F.conv2d(xr, q(w1), b1))) < tol
    got = ops.conv3x3s2_bias_act(xd, w3.to(DEV), b3.to(DEV), act)
    assert got.shape == (2, 48, 6, 7) and _rel(got, ACTS[act](F.conv2d(xr, q(w3), b3, 2, 1))) < tol
    if act not in (torch_ops.ACT_NONE, torch_ops.ACT_SILU):   # the depth-wise kernel has the two epilogues the graph uses (heads: none, bottlenecks: SiLU)
        with pytest.raises(RuntimeError):
            ops.dwconv_bias_act(xd, wd.to(DEV), bd.to(DEV), act)
        return
    got = ops.dwconv_bias_act(xd, wd.to(DEV), bd.to(DEV), act)
    assert _rel(got, ACTS[act](F.conv2d(xr, q(wd), bd, 1, 3, 1, 64))) < tol
    # NCHW-contiguous input (what a reference caller hands over) gives the same result
    assert torch.equal(ops.dwconv_bias_act(x.to(DEV), wd.to(DEV), bd.to(DEV), act), got)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float16, 3e-2)])
def test_autograd_through_the_ops(ops, dtype, tol):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 48, 12, 10, generator=g)
    w1, b1 = torch.randn(72, 48, 1, 1, generator=g) / 7, torch.randn(72, generator=g)
    w3 = torch.randn(64, 72, 3, 3, generator=g) / 25
    wd = torch.randn(64, 1, 5, 5, generator=g) / 5
    q = (lambda t: t.detach().clone().to(dtype).float())

    def net(x_, w1_, b1_, w3_, wd_, f1, f3, fd):
        return fd(F.relu(f3(F.silu(f1(x_, w1_, b1_)), w3_)), wd_)
    rx = q(x).requires_grad_(True); r1 = w1.clone().requires_grad_(True); rb = b1.clone().requires_grad_(True); r3 = w3.clone().requires_grad_(True); rd = wd.clone().requires_grad_(True)
    qw = (lambda t: t.to(dtype).float())                     # differentiable rounding of the (cloned) master weights
    ref = net(rx, r1, rb, r3, rd, lambda a, w_, b_: F.conv2d(a, qw(w_), b_), lambda a, w_: F.conv2d(a, qw(w_), None, 2, 1), lambda a, w_: F.conv2d(a, qw(w_), None, 1, 2, 1, 64))
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(q(dy))
    gx = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g1, gb, g3, gd = (t.to(DEV).requires_grad_(True) for t in (w1, b1, w3, wd))
    out = net(gx, g1, gb, g3, gd, lambda a, w_, b_: ops.conv1x1_bias_act(a, w_, b_, 0), lambda a, w_: ops.conv3x3s2_bias_act(a, w_, None, 0), lambda a, w_: ops.dwconv_bias_act(a, w_, None, 0))
    out.backward(dy.to(DEV).to(dtype))
    assert _rel(out, ref.detach()) < tol
    for got, want, name in ((gx.grad, rx.grad, "dx"), (g1.grad, r1.grad, "dw1"), (gb.grad, rb.grad, "db1"), (g3.grad, r3.grad, "dw3"), (gd.grad, rd.grad, "dwd")):
        assert got is not None and _rel(got, want) < tol, name
    with pytest.raises(RuntimeError):                         # a fused activation has no registered backward
        ops.conv1x1_bias_act(gx, g1, gb, torch_ops.ACT_SILU)


def test_autocast_rule_casts_like_a_convolution(ops):
    x = torch.randn(1, 32, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(32, 32, 1, 1, device=DEV)
    assert ops.conv1x1_bias_act(x, w, None, 0).dtype == torch.float32
    with torch.autocast("cuda", dtype=torch.float16):
        y = ops.conv1x1_bias_act(x, w, None, 0)
        z = ops.conv3x3s2_bias_act(x, torch.randn(16, 32, 3, 3, device=DEV), None, 0)
    assert y.dtype == torch.float16 and z.dtype == torch.float16
    assert _rel(y, F.conv2d(x.cpu().half().float(), w.cpu().half().float())) < 4e-3


def test_head_decode_and_decode_nms_match_the_oracle_and_the_ctypes_path(ops):
    g = torch.Generator().manual_seed(6)
    B, dims = 2, [(12, 20), (6, 10), (3, 5)]
    heads = [(torch.zeros(B, 1, h, w), torch.sigmoid(torch.randn(B, 80, h, w, generator=g) * 2 - 3), torch.randn(B, 68, h, w, generator=g) * 2) for h, w in dims]
    ref = O.decode(heads)
    pred = ops.head_decode([h[1].to(DEV) for h in heads], [h[2].to(DEV).half() for h in heads], [8.0, 16.0, 32.0])
    np.testing.assert_allclose(pred[..., :4].cpu().numpy(), O.decode([(a, b_, c.half().float()) for a, b_, c in heads])[..., :4].numpy(), rtol=1e-5, atol=1e-3)
    assert torch.equal(pred[..., 4:].cpu(), ref[..., 4:])
    dets = torch_ops.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
    want = M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
    odets = O.non_max_suppression(pred.cpu().numpy(), 0.03, 0.65, multi_label=True)
    assert len(dets) == B and all(torch.equal(a, b_) for a, b_ in zip(dets, want)) and all(np.array_equal(a.cpu().numpy(), b_) for a, b_ in zip(dets, odets))
    sel = torch_ops.non_max_suppression(pred, 0.03, 0.65, classes=[3, 17], agnostic=True, max_det=50)
    osel = O.non_max_suppression(pred.cpu().numpy(), 0.03, 0.65, classes=[3, 17], agnostic=True, max_det=50)
    assert all(np.array_equal(a.cpu().numpy(), b_) for a, b_ in zip(sel, osel))
    with pytest.raises(AssertionError):
        torch_ops.non_max_suppression(pred, 1.5, 0.65)
    with pytest.raises(RuntimeError):                         # the op itself reports bad thresholds as RuntimeError (TORCH_CHECK)
        ops.decode_nms(pred, 0.03, -0.1, False, False, 300, None)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float16, 5e-3)])
def test_mprep_and_sppf_ops(ops, dtype, tol):
    """SURVEY.md 8(b) op list: MPRep (common.py:776-792) and SPPF (:114-129) in deploy form as single ops against their torch definitions."""
    g = torch.Generator().manual_seed(21)
    q = (lambda t: t.half().float()) if dtype == torch.float16 else (lambda t: t)
    x = torch.randn(2, 48, 20, 24, generator=g).to(dtype)
    w1, b1 = torch.randn(32, 48, 1, 1, generator=g) / 7, torch.randn(32, generator=g)
    w3, b3 = torch.randn(32, 48, 3, 3, generator=g) / 20, torch.randn(32, generator=g)
    xr = x.float()
    ref = torch.cat([F.silu(F.conv2d(F.max_pool2d(xr, 2, 2), q(w1), b1)), F.relu(F.conv2d(xr, q(w3), b3, 2, 1))], 1)
    got = ops.mprep(x.to(DEV), w1.to(DEV).to(dtype), b1.to(DEV), w3.to(DEV).to(dtype), b3.to(DEV))
    assert got.shape == ref.shape and got.dtype == dtype and _rel(got, ref) < tol
    v1, c1 = torch.randn(24, 48, 1, 1, generator=g) / 7, torch.randn(24, generator=g)
    v2, c2 = torch.randn(64, 96, 1, 1, generator=g) / 10, torch.randn(64, generator=g)
    y = F.silu(F.conv2d(xr, q(v1), c1))
    if dtype == torch.float16:
        y = y.half().float()                         # cv1's output is stored in fp16 before the pools and cv2 read it
    y1 = F.max_pool2d(y, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    ref = F.silu(F.conv2d(torch.cat([y, y1, y2, y3], 1), q(v2), c2))
    got = ops.sppf(x.to(DEV), v1.to(DEV).to(dtype), c1.to(DEV), v2.to(DEV).to(dtype), c2.to(DEV))
    assert got.shape == ref.shape and _rel(got, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 1e-2)])
@pytest.mark.parametrize("act,name", [(torch_ops.ACT_NONE, None), (torch_ops.ACT_SILU, "silu"), (torch_ops.ACT_RELU, "relu")])
def test_bn_act_op_forward_backward(ops, dtype, tol, act, name):
    """BatchNorm2d(train) + activation (Conv.forward, common.py:44-47) as an op with autograd: y, running statistics, dx, dgamma, dbeta vs torch."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 48, 9, 11, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(48, generator=g) + 0.5, torch.randn(48, generator=g)
    dy = torch.randn(3, 48, 9, 11, generator=g)
    bn = torch.nn.BatchNorm2d(48, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    yr = bn(xr)
    yr = yr if name is None else (F.silu(yr) if name == "silu" else F.relu(yr))
    yr.backward(dy)
    xd = x.detach().to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).clone().requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(48, device=DEV), torch.ones(48, device=DEV)
    y, mean, rstd, rm, rv = ops.bn_act(xd, gd, bd, rm, rv, 1e-3, 0.03, act)
    y.backward(dy.to(DEV).to(dtype))
    assert _rel(y, yr) < tol and _rel(xd.grad, xr.grad) < tol * 3
    assert _rel(gd.grad, bn.weight.grad) < tol * 3 and _rel(bd.grad, bn.bias.grad) < tol * 3
    assert _rel(rm, bn.running_mean) < 1e-3 and _rel(rv, bn.running_var) < 1e-3


@pytest.mark.parametrize("scale", ["n", "s"])
def test_model_dispatch_ops_equals_the_engine_and_traces(scale):
    """`Model(dispatch="ops")`: the eval forward as a sequence of torch.ops.mafyolo.* calls (ops_forward.py) gives the engine's prediction (fp32: the
    same kernels in the unfused plan; fp16: against the fused plan within the fp16 bars) and traces under torch.compile(fullgraph=True) through the
    fake kernels — the dispatcher-visible form of the hot path."""
    from maf_yolo_amd import synth
    eng = M.Model(scale)
    eng.load_state_dict(synth.synth_state_dict(eng, scale, 0))
    eng = eng.to(DEV).eval()
    opsm = M.Model(scale, dispatch="ops")
    opsm.load_state_dict(eng.state_dict())
    opsm = opsm.to(DEV).eval()
    x = synth.synth_images(2, 128, seed=3).to(DEV)
    with torch.no_grad():
        ref = eng(x)[0]
        got, heads = opsm(x)
        assert got.shape == ref.shape and len(heads) == 3
        d = (got - ref).abs()
        assert d[..., :4].max() <= 1e-3 + 1e-5 * ref[..., :4].abs().max() and d[..., 4:].max() <= 2e-5, (d[..., :4].max(), d[..., 4:].max())
        xh = x.half()
        refh, goth = eng(xh)[0], opsm(xh)[0]
        dh = (goth - refh).abs()
        assert dh[..., :4].max() <= 1.0 and dh[..., 4:].max() <= 1e-2, (dh[..., :4].max(), dh[..., 4:].max())
        f = opsm.traceable(torch.float32)
        compiled = torch.compile(f, backend="eager", fullgraph=True)
        xin = x.contiguous(memory_format=torch.channels_last)
        assert torch.equal(compiled(xin), got) and torch.equal(f(xin), got)
