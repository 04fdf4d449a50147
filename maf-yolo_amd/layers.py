"""Train-form blocks of MAF-YOLO with the reference's parameter names, plus their deploy switch.

The module tree mirrors the attribute names of yolov6/layers/common.py (RepVGGBlock :166, Conv :29,
ConvWrapper :76, SPPF :114, MPRep :776, DepthBottleneckUni :898, RepHDW :928, Head_DepthUni :1288,
DilatedReparamBlock :2948, UniRepLKNetBlock :3053) so a reference state_dict (838 / 1206 / 1568
tensors for n / s / m) loads with strict=True and trained weights round-trip.

`forward` here is the TRAINING-form graph: every convolution (1x1, depth-wise, 3x3 stride-2 and the stride-2 1x1 of the
RepVGG blocks) and every BatchNorm(train)+activation runs forward and backward on the HIP kernels through train_ops.py;
what is left to torch is element-wise glue on channels_last tensors (branch sums, max-pool, cat, upsample).
Inference never runs these forwards: in eval mode
Model.forward executes the re-parameterised graph on the HIP engine (engine.py), built from
`fused()` below — the deploy algebra of SURVEY.md §3.3, evaluated in fp32 on the host once.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .arch import dil_branch_kernels
from . import train_ops

BN_EPS, BN_MOMENTUM = 1e-3, 0.03          # yolov6/utils/torch_utils.py:43-45


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOMENTUM)


def fold_bn(weight, bn, bias=None):
    """(conv weight, BN) -> equivalent (weight, bias): w*g/s, b' = beta + (b - mean)*g/s  with s = sqrt(var+eps)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    b0 = bn.running_mean.new_zeros(bn.running_mean.shape) if bias is None else bias
    return weight * scale.reshape(-1, 1, 1, 1), bn.bias + (b0 - bn.running_mean) * scale


class ConvBN(nn.Sequential):
    """conv (no bias) + BN pair named `.conv` / `.bn` (the reference's conv_bn helper, common.py:157-163)."""

    def __init__(self, cin, cout, k, stride, pad, groups=1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=False))
        self.add_module("bn", _bn(cout))

    def fused(self):
        return fold_bn(self.conv.weight, self.bn)


class Conv(nn.Module):
    """k x k conv + BN + SiLU (common.py:29-50)."""

    def __init__(self, cin, cout, k=1, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
        self.bn = _bn(cout)
        self.act = nn.SiLU(inplace=True)

    def forward(self, x, out=None):
        """out: where the result goes when it is one input of a channel concat — a slot of a train_ops.CatBuffer, or a callable that returns the slot for the
        conv's output tensor (the buffer is allocated when its first producer knows the spatial size); ignored off the HIP path (the concat then copies)."""
        pre = None
        if self.conv.kernel_size == (1, 1) and self.conv.stride == (1, 1):
            z, pre = train_ops.conv1x1_bn(x, self.conv.weight, self.bn)         # HIP conv fwd / dgrad / wgrad; pre: the BatchNorm's batch statistics out of the conv's epilogue
        elif self.conv.kernel_size == (3, 3) and self.conv.stride == (2, 2):
            z = train_ops.conv3x3s2(x, self.conv.weight)                        # ConvWrapper: the same on the 3x3 stride-2 kernels
        else:
            z = self.conv(x)
        if out is not None:
            out = (out(z) if callable(out) else out) if train_ops.cat_free_ok(z, self.bn) and z.dim() == 4 and z.shape[1] % 8 == 0 else None
        return train_ops.bn_act(z, self.bn, "silu", out=out, pre_stats=pre)

    def fused(self):
        return fold_bn(self.conv.weight, self.bn)


class ConvWrapper(nn.Module):
    def __init__(self, cin, cout, k=3, stride=2):
        super().__init__()
        self.block = Conv(cin, cout, k, stride)

    def forward(self, x, out=None):
        return self.block(x, out=out)


class RepVGGBlock(nn.Module):
    """3x3 s2 conv+BN  +  1x1 s2 conv+BN, ReLU (common.py:219-224). No identity branch exists at stride 2."""

    def __init__(self, cin, cout):
        super().__init__()
        self.rbr_dense = ConvBN(cin, cout, 3, 2, 1)
        self.rbr_1x1 = ConvBN(cin, cout, 1, 2, 0)
        self.nonlinearity = nn.ReLU(inplace=True)

    def forward(self, x, out=None):
        # both branches (conv + BatchNorm each) on the HIP kernels
        if x.is_cuda and x.shape[1] % 8:
            x = train_ops.pad_channels8(x)                                       # the image: cast + channel padding once for both branches
        if out is not None and not (train_ops.cat_free_ok(x, self.rbr_dense.bn) and train_ops.cat_free_ok(x, self.rbr_1x1.bn) and self.rbr_dense.conv.out_channels % 8 == 0):
            out = None                                                           # (a frozen branch BatchNorm: the torch path cannot store into a slot — the concat copies, as in Conv.forward)
        elif callable(out):
            out = None                                                           # (a slot that does not exist yet needs the output's shape first: only Conv resolves those)
        # ReLU(BN(3x3) + BN(1x1)): one apply pass over both branch tensors (csrc/bn_sum.hip; backward: one statistics + one apply launch for both BatchNorms,
        # the ReLU's mask recomputed from the branch tensors)
        # (the two convs are one autograd node: the 1x1 branch's data gradient is added onto the 3x3 branch's on the even pixels, in place)
        return train_ops.bn_sum(list(train_ops.repvgg_convs(x, self.rbr_dense.conv.weight, self.rbr_1x1.conv.weight)),
                                [self.rbr_dense.bn, self.rbr_1x1.bn], act="relu", out=out)

    def fused(self):
        """One 3x3 kernel + bias (get_equivalent_kernel_bias, common.py:226-230)."""
        w3, b3 = self.rbr_dense.fused()
        w1, b1 = self.rbr_1x1.fused()
        return w3 + F.pad(w1, [1, 1, 1, 1]), b3 + b1

    def switch_to_deploy(self):      # evaler.py:101-103 calls this on every RepVGGBlock: harmless here
        return None


class DilatedReparamBlock(nn.Module):
    """Depth-wise k x k conv+BN plus parallel smaller depth-wise conv+BN branches (common.py:2948-3031)."""

    def __init__(self, c, k):
        super().__init__()
        self.kernel_size = k
        self.kernel_sizes = list(dil_branch_kernels(k))
        self.lk_origin = nn.Conv2d(c, c, k, 1, k // 2, groups=c, bias=False)
        self.origin_bn = _bn(c)
        for kk in self.kernel_sizes:
            setattr(self, "dil_conv_k%d_1" % kk, nn.Conv2d(c, c, kk, 1, kk // 2, groups=c, bias=False))
            setattr(self, "dil_bn_k%d_1" % kk, _bn(c))

    def forward(self, x, next_bn=None):
        # every branch in ONE launch (csrc/dw_branches.hip: x staged once, the 1 x 1 scale branch rides along; their data gradients summed in one launch
        # too), which also accumulates every branch's BatchNorm statistics in its epilogue; the BatchNorms of all branches are summed by ONE apply pass
        # (csrc/bn_sum.hip; backward: one statistics + one apply launch for all of them)
        names = ["lk_origin"] + ["dil_conv_k%d_1" % kk for kk in self.kernel_sizes]
        bns = [self.origin_bn] + [getattr(self, "dil_bn_k%d_1" % kk) for kk in self.kernel_sizes]
        zz, pre = train_ops.dw_branches(x, [getattr(self, n_).weight for n_ in names], bns)
        # (next_bn: the BatchNorm behind the block — UniRepLKNetBlock.norm —, whose batch statistics the sum's apply pass accumulates: returns (sum, pre_stats) then)
        return train_ops.bn_sum(zz, bns, pre, next_bn=next_bn)                                        # origin_bn(z_0) + sum_j dil_bn_j(z_j)

    def fused(self):
        """merge_dilated_branches (common.py:3033-3051): centre-pad each small kernel to k x k and sum."""
        k = self.kernel_size
        w, b = fold_bn(self.lk_origin.weight, self.origin_bn)
        for kk in self.kernel_sizes:
            bw, bb = fold_bn(getattr(self, "dil_conv_k%d_1" % kk).weight, getattr(self, "dil_bn_k%d_1" % kk))
            p = k // 2 - kk // 2
            w = w + F.pad(bw, [p, p, p, p])
            b = b + bb
        return w, b


class UniRepLKNetBlock(nn.Module):
    """DilatedReparamBlock followed by an outer BN (common.py:3053-3083)."""

    def __init__(self, c, k):
        super().__init__()
        self.dwconv = DilatedReparamBlock(c, k)
        self.norm = _bn(c)

    def forward(self, x, act=None):
        if x.is_cuda and self.norm.training:
            y, pre = self.dwconv(x, next_bn=self.norm)                          # the norm's statistics out of the branch sum's apply pass (csrc/bn_sum.hip)
            return train_ops.bn_act(y, self.norm, act, pre_stats=pre)
        return train_ops.bn_act(self.dwconv(x), self.norm, act)               # `act`: the SiLU of DepthBottleneckUni fused into the norm

    def fused(self):
        """reparameterize (common.py:3085-3100): fold the outer BN into the merged depth-wise kernel."""
        w, b = self.dwconv.fused()
        return fold_bn(w, self.norm, b)

    def reparameterize(self):        # evaler.py:107-109 calls this on every UniRepLKNetBlock: harmless here
        return None


class DepthBottleneckUni(nn.Module):
    """1x1 (c -> 3c) -> depth-wise k x k -> SiLU -> 1x1 (3c -> c)   (common.py:898-927)."""

    def __init__(self, c, k, expansion=3):
        super().__init__()
        mid = int(c * expansion)
        self.conv1 = Conv(c, mid, 1)
        self.conv2 = UniRepLKNetBlock(mid, k)
        self.act = nn.SiLU(inplace=True)
        self.one_conv = Conv(mid, c, 1)

    def forward(self, x, out=None):
        return self.one_conv(self.conv2(self.conv1(x), act="silu"), out=out)


class RepHDW(nn.Module):
    """conv1 1x1 -> split -> chained DepthBottleneckUni -> cat -> conv2 1x1   (common.py:928-946)."""

    def __init__(self, cin, cout, depth=1, expansion=0.5, k=5, depth_expansion=3):
        super().__init__()
        self.c_ = int(cout * expansion)
        self.conv1 = Conv(cin, 2 * self.c_, 1)
        self.m = nn.ModuleList(DepthBottleneckUni(self.c_, k, depth_expansion) for _ in range(depth))
        self.conv2 = Conv(self.c_ * (depth + 2), cout, 1)

    def forward(self, x, out=None):
        c = self.c_
        if train_ops.cat_free_ok(x, self.conv1.bn) and c % 8 == 0:
            # no cat, no split: conv1's and every block's last BatchNorm apply pass store into their slots of one buffer (train_ops.CatBuffer)
            z = train_ops.conv1x1(x, self.conv1.conv.weight)
            cb = train_ops.CatBuffer(z, [c] * (len(self.m) + 2))
            t = train_ops.bn_act(z, self.conv1.bn, "silu", out=cb.slot(0, 2))
            part, cur = train_ops.fork(t, c)                                     # both halves go to conv2, the second one also into the first block
            parts = [part]
            for i, blk in enumerate(self.m):
                y = blk(cur, out=cb.slot(2 + i))
                if i + 1 < len(self.m):
                    y, cur = train_ops.fork(y)
                parts.append(y)
            return self.conv2(train_ops.join(cb, parts), out=out)
        if x.is_cuda:
            train_ops._glue()                                                    # (split / cat as torch kernels: not a step a tape may record)
        t = self.conv1(x)
        outs = list(t.split((c, c), 1))                                          # split, not slices: its backward is ONE cat of the two gradients
        for blk in self.m:
            outs.append(blk(outs[-1]))
        return self.conv2(torch.cat(outs, 1), out=out)


class MP(nn.Module):
    def __init__(self):
        super().__init__()
        self.m = nn.MaxPool2d(2, 2)

    def forward(self, x):
        return train_ops.maxpool(x, 2, 2, 0)                                     # csrc/pool_train.hip on CUDA tensors, else F.max_pool2d


class MPRep(nn.Module):
    """cat[Conv1x1(maxpool2x2(x)), RepVGG3x3s2(x)]   (common.py:776-792)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.mp = MP()
        self.conv1 = Conv(cin, cout // 2, 1)
        self.conv2 = RepVGGBlock(cin, cout // 2)

    def forward(self, x):
        if train_ops.cat_free_ok(x, self.conv1.bn) and self.conv1.conv.out_channels % 8 == 0:
            c = self.conv1.conv.out_channels                                     # both halves store into their slots of one buffer (train_ops.CatBuffer): no cat
            holder = []

            def slot0(z):
                holder.append(train_ops.CatBuffer(z, [c, c]))
                return holder[0].slot(0)

            xa, xb = train_ops.fanout(x, 2)                                      # two readers: their gradients meet in one launch (train_ops._Fanout)
            # (the pooled branch on a lane of a recording step tape, beside the 3 x 3 branch on the main stream)
            a, lane = train_ops.lane_run(1, lambda t: self.conv1(self.mp(t), out=slot0), xa)
            if holder and train_ops.cat_free_ok(x, self.conv2.rbr_dense.bn) and train_ops.cat_free_ok(x, self.conv2.rbr_1x1.bn):
                b = self.conv2(xb, out=holder[0].slot(1))
                return train_ops.join(holder[0], [train_ops.lane_join(a, lane), b])
            if x.is_cuda:
                train_ops._glue()
            return torch.cat([train_ops.lane_join(a, lane), self.conv2(xb)], 1)
        if x.is_cuda:
            train_ops._glue()
        return torch.cat([self.conv1(self.mp(x)), self.conv2(x)], 1)


class SPPF(nn.Module):
    """cv1 1x1 -> three chained 5x5 max pools -> cat -> cv2 1x1   (common.py:114-129)."""

    def __init__(self, cin, cout, k=5):
        super().__init__()
        c_ = cin // 2
        self.cv1 = Conv(cin, c_, 1)
        self.cv2 = Conv(c_ * 4, cout, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x, out=None):
        k = self.m.kernel_size
        c = self.cv1.conv.out_channels
        if train_ops.cat_free_ok(x, self.cv1.bn) and c % 8 == 0 and x.dtype in (torch.float16, torch.float32) and 2 <= k <= 15:
            # no cat: cv1's apply pass and the three pooling kernels store into their slots of one buffer; a map that feeds the concat AND the next pool gets the
            # pool's gradient added into its slot of the concat's gradient (train_ops.fork)
            holder = []

            def slot0(z):
                holder.append(train_ops.CatBuffer(z, [c] * 4))
                return holder[0].slot(0)

            t = self.cv1(x, out=slot0)
            if holder and train_ops.maxpool_native_ok(t, k):
                cb = holder[0]
                parts = []
                for i in range(3):
                    keep, nxt = train_ops.fork(t)
                    parts.append(keep)
                    t = train_ops.maxpool_s1(nxt, k, out=cb.slot(i + 1))
                parts.append(t)
                return self.cv2(train_ops.join(cb, parts), out=out)
            x = t
        else:
            x = self.cv1(x)
        if x.is_cuda:
            train_ops._glue()
        y1 = train_ops.maxpool_s1(x, k)                                          # csrc/pool_train.hip on CUDA tensors (gather backward), else F.max_pool2d
        y2 = train_ops.maxpool_s1(y1, k)
        return self.cv2(torch.cat((x, y1, y2, train_ops.maxpool_s1(y2, k)), 1), out=out)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, xs):
        if xs[0].is_cuda:
            train_ops._glue()
        return torch.cat(xs, self.d)


class Head_DepthUni(nn.Module):
    """Decoupled depth-wise head: stem 1x1; per branch UniRepLK k x k -> 1x1 -> pred 1x1   (common.py:1288-1336)."""

    def __init__(self, cin, c, reg_max=16, k=5, nc=80):
        super().__init__()
        self.stem = Conv(cin, c, 1)
        self.cls_conv = UniRepLKNetBlock(c, k)
        self.cls_conv_s = Conv(c, c, 1)
        self.reg_conv = UniRepLKNetBlock(c, k)
        self.reg_conv_s = Conv(c, c, 1)
        self.cls_pred = nn.Conv2d(c, nc, 1)
        self.reg_pred = nn.Conv2d(c, 4 * (reg_max + 1), 1)
        # common.py:1307-1323: zero pred weights, cls bias = -log((1-p)/p) with p = 0.01, reg bias = 1
        with torch.no_grad():
            self.cls_pred.weight.zero_()
            self.cls_pred.bias.fill_(-math.log((1 - 1e-2) / 1e-2))
            self.reg_pred.weight.zero_()
            self.reg_pred.bias.fill_(1.0)

    def forward(self, x, raw=False, lanes=None):
        """(stem features, class probabilities, box distributions); raw=True: the class LOGITS (Model applies the sigmoid behind a step tape's boundary).
        lanes = (a, b): the two branches on lanes a and b of a recording step tape (train_ops.lane_run) — then (stem, logits, box, lane of the logits, lane of the
        box tensor) comes back and the caller joins the lanes (train_ops.lane_join) before the main stream reads the two tensors."""
        x = self.stem(x)
        xc, xr = train_ops.fanout(x, 2)                                          # two branches read the stem: one launch sums their gradients

        def cls_branch(t):
            return train_ops.conv1x1(self.cls_conv_s(self.cls_conv(t)), self.cls_pred.weight, self.cls_pred.bias)

        def reg_branch(t):
            return train_ops.conv1x1(self.reg_conv_s(self.reg_conv(t)), self.reg_pred.weight, self.reg_pred.bias)

        if lanes is not None:
            cls, kc = train_ops.lane_run(lanes[0], cls_branch, xc)
            reg, kr = train_ops.lane_run(lanes[1], reg_branch, xr)
            return x, (cls if raw else torch.sigmoid(cls)), reg, kc, kr
        cls, reg = cls_branch(xc), reg_branch(xr)
        return x, (cls if raw else torch.sigmoid(cls)), reg


class Out(nn.Module):
    def forward(self, xs):
        return list(xs)
