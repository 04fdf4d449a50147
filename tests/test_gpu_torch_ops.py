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
