"""Static launch plan for the re-parameterised (deploy-form) graph on one GPU.

Model.forward of the reference walks 35 nn.Modules and issues ~210 aten ops per call
(yolov6/models/yolo.py:186-201).  Here the same graph is flattened ONCE, for a
given (batch, H, W, dtype), into an ordered list of `maf_op_t` launches over a single activation
arena (one torch uint8 tensor = device memory plumbing only) and handed to the C engine
(`maf_engine_create`); a forward is then one C call.  Concat, split, nearest-upsample and the 2x2
max-pool of MPRep never become ops: they turn into operand addressing of the consumer conv
(`TV` segments below), and SPPF's three max-pools write straight into the concat buffer.
"""
import ctypes as C
import os

import torch

from .config import cfg
from . import lib, pack
from .layers import (RepVGGBlock, RepHDW, MPRep, SPPF, ConvWrapper, Head_DepthUni)

_ESIZE = {lib.F16: 2, lib.F32: 4}


from .tuner import _TUNE_CACHE, choose_fusion, stream_lds_ok, save_tune_cache, load_tune_cache, p2_wave_bytes      # noqa: F401  (measured choices: tuner.py)
from . import tuner as _tuner, plan_report as _report

_TORCH_DT = {lib.F16: torch.float16, lib.F32: torch.float32}


class Buf:
    """A real NHWC buffer in the arena."""

    def __init__(self, off, H, W, stride, esize, C=None):
        self.off, self.H, self.W, self.stride, self.esize = off, H, W, stride, esize
        self.C = stride if C is None else C       # logical channels (< stride when the conv's Cout was padded, see cls_pred)


class Seg:
    def __init__(self, buf, C, coff=0, mode=lib.SRC_DIRECT):
        self.buf, self.C, self.coff, self.mode = buf, C, coff, mode


class TV:
    """Virtual tensor = channel-concatenation of segments on an H x W grid."""

    def __init__(self, segs, H, W):
        self.segs, self.H, self.W = segs, H, W

    @property
    def C(self):
        return sum(s.C for s in self.segs)


class Plan:
    def __init__(self, model, B, Hin, Win, dtype, in_dtype, device, fuse=None, fuse_head=None, fuse_tail=None):
        assert Hin % 32 == 0 and Win % 32 == 0, "image sides must be multiples of 32 (stride of P5)"
        self.B, self.Hin, self.Win, self.dtype, self.in_dtype, self.device = B, Hin, Win, dtype, in_dtype, device
        self.es = _ESIZE[dtype]
        self._arena_size = 0
        self._wblobs = []          # (offset, cpu tensor)
        self._wsize = 0
        self._ops = []             # python-side op records (dicts), turned into MafOp once addresses are known
        self.nc = model.nc
        self.reg_max = model.detect.reg_max
        self.strides = [float(s) for s in model.detect.stride.tolist()]
        # A DepthBottleneckUni runs as three launches (mode 0), as ONE launch (mode 1, csrc/bottleneck.hip, c <= 64) or as conv1 +
        # depth-wise fused followed by the plain 1x1 (mode 2, csrc/conv1dw.hip, any c).  `fuse`: False (0 everywhere), True (1 where
        # it exists, else 2), 2, a {name: mode} dict, or "auto" = the built-in rule (mode 1 for k <= 5 and c <= 64: the 160^2 / 80^2
        # maps, where it wins by 1.3-1.8x; Model.plan_for replaces the rule by a measurement when autotuning is on).
        self.fuse = getattr(model, "fuse_bottlenecks", "auto") if fuse is None else fuse
        # independent branches (side down-sampling convs of the MAFPN neck, the three heads and their cls / reg halves) CAN run
        # on separate HIP streams of the engine.  Measured on MI355X (n, bs 32): heads-only lanes give -1 % on forward+NMS and
        # +1.3 % on forward alone (cross-stream event latency eats the overlap), all lanes lose 1-2 %: opt-in.
        # the tail of every detection level ({cls,reg}_conv_s -> {cls,reg}_pred -> sigmoid / DFL decode) as one launch per level
        # (csrc/head_tail.hip, fp16, 80 classes, head width 64 / 128 / 192 and — weights streamed from L2 — 256 / 384: every level of n, s, m)
        # instead of four 1x1 convs + the decode kernel
        fh = getattr(model, "fuse_head", "auto") if fuse_head is None else fuse_head
        self.fuse_head = bool(fh) and dtype == lib.F16 and model.nc == 80 and model.detect.reg_max == 16
        fs = getattr(model, "fuse_stem", True)                   # False | 1: backbone.0 + backbone.1 | True / 2: + the 1x1 that opens backbone.2
        self.fuse_stem = (2 if fs is True else int(fs or 0)) if dtype == lib.F16 else 0
        self._stem2 = self._stem3 = None
        # MPRep's two branches in one launch where the kernel exists: True / False / "auto" = in tuned plans only (the untuned default plan keeps one
        # launch list for every batch size, so that an image's rows do not depend on how many images ran beside it — the fused kernel sums in another order)
        fm = getattr(model, "fuse_mprep", "auto")
        self.fuse_mprep = (bool(getattr(model, "autotune", False)) if fm == "auto" else bool(fm)) and cfg.fuse_mprep
        # the conv that closes a RepHDW block inside the launch of its last (fully fused) bottleneck where that instantiation exists (csrc/bottleneck.hip, op.nc):
        #   "auto" (default)  blocks of ONE bottleneck: every block of n.  The end-to-end detection bars of s / m (tests/test_gpu_fused_parity.py) are the ones measured
        #                     on this setting.
        #   True              every block whose instantiation exists: + the first two blocks of s and the first of m (two bottlenecks).  Opt-in (`model.fuse_tail = True`, or
        #                     MAF_FUSE_TAIL=3 for "auto" models): on s one pair whose IoU sits at the NMS threshold (0.65010 against 0.65) falls the other way and the survivor
        #                     suppresses five neighbours (profiles/round5_fuse_tail_flip_s.txt) — scores within 1.5e-3 either way, but the 640^2 check then matches 289 rows
        #                     instead of 293; the opt-in has its own, wider, bars in the test and is not what bench.py times.
        #   False             off.   MAF_FUSE_TAIL: 0 = off for every model, 1 = "auto" rule (the default), 3 = "auto" means True.
        ft = getattr(model, "fuse_tail", "auto") if fuse_tail is None else fuse_tail
        env_ft = str(cfg.fuse_tail)
        deep = 3 if env_ft == "3" else 1
        self.fuse_tail = (deep if ft == "auto" else 3 if ft else 0) if (env_ft != "0" and dtype == lib.F16) else 0      # deepest block (bottlenecks) taken
        self.split_cat = bool(getattr(model, "split_cat", cfg.split_cat))   # RepHDW behind the fused stem: one dense tensor per concat slot (see the rephdw branch)
        self.lanes = getattr(model, "multi_stream", False)
        if isinstance(self.lanes, bool):
            self.lanes = 2 if self.lanes else 0                  # 0: one stream; 1: heads only; 2: heads + neck side convs
        with torch.no_grad():
            self._build(model)
        self._finalize()

    def _fuse_mode(self, name, k, c):
        if self.dtype != lib.F16 or c % 8:
            return 0
        f = self.fuse
        if isinstance(f, dict):
            m = f.get(name, 0)
        elif isinstance(f, (set, frozenset, list, tuple)):
            m = 1 if name in f else 0
        elif f == "auto":
            m = 1 if k <= 5 else 0
        elif f is True:
            m = 1
        else:
            m = int(f or 0)
        if m == 1 and c > 64:
            m = 2 if (f is True or isinstance(f, dict)) else 0
        if m == 2 and c > 512:
            m = 0
        return m

    # ---------------------------------------------------------------- allocation helpers
    def _alloc(self, H, W, C, esize=None):
        esize = esize or self.es
        off = self._arena_size
        self._arena_size += (self.B * H * W * C * esize + 255) // 256 * 256
        return Buf(off, H, W, C, esize)

    def _wput(self, t):
        t = t.contiguous()
        off = self._wsize
        self._wsize += (t.numel() * t.element_size() + 255) // 256 * 256
        self._wblobs.append((off, t))
        return off

    # ---------------------------------------------------------------- op emitters
    def _conv1x1(self, name, w, b, src, out, out_coff, act, out_f32=False):
        cout = w.shape[0]
        M = self.B * src.H * src.W
        pt, ct = pack.tile_for(cout, M)
        srcC = [s.C for s in src.segs]
        assert len(src.segs) <= 4, "%s: more than 4 concat sources" % name
        self._ops.append(dict(kind=lib.OP_CONV1X1, name=name, act=act, H=src.H, W=src.W, Cin=src.C, Cout=cout, raw=(w.detach().float().cpu(), b.detach().float().cpu(), srcC),
                              segs=src.segs, out=out, out_coff=out_coff, out_f32=int(out_f32), pt=pt, ct=ct,
                              w=self._wput(pack.pack_conv1x1(w, srcC, ct, self.dtype)), b=self._wput(pack.pack_bias(b, ct))))

    def _conv3x3s2(self, name, w, b, src, out, out_coff, act, twin=None):
        """twin = (name2, w2, b2, src2, out2): a second, independent conv of the same shape launched as blockIdx.y = 1 of the same grid."""
        assert len(src.segs) == 1 and src.segs[0].mode == lib.SRC_DIRECT, "%s: 3x3 s2 conv needs a materialised input" % name
        cout = w.shape[0]
        H, W = src.H // 2, src.W // 2
        pt, ct = pack.tile_for(cout, self.B * H * W)
        rec = dict(kind=lib.OP_CONV3X3S2, name=name, act=act, H=H, W=W, Hin=src.H, Win=src.W, Cin=src.C, Cout=cout, raw=(w.detach().float().cpu(), b.detach().float().cpu(), None),
                   segs=src.segs, out=out, out_coff=out_coff, out_f32=0, pt=pt, ct=ct,
                   w=self._wput(pack.pack_conv3x3(w, ct, self.dtype)), b=self._wput(pack.pack_bias(b, ct)))
        if twin is not None:
            name2, w2, b2, src2, out2 = twin
            s1, s2 = src.segs[0], src2.segs[0]
            assert len(src2.segs) == 1 and s2.mode == lib.SRC_DIRECT and (s2.buf.stride, s2.coff, s2.C) == (s1.buf.stride, s1.coff, s1.C) and w2.shape == w.shape \
                and (out2.stride, out2.H, out2.W) == (out.stride, out.H, out.W) and out_coff == 0
            rec.update(name=name + "+" + name2, twin=dict(seg=s2, out=out2, raw=(w2.detach().float().cpu(), b2.detach().float().cpu()),
                                                          w=self._wput(pack.pack_conv3x3(w2, ct, self.dtype)), b=self._wput(pack.pack_bias(b2, ct))))
        self._ops.append(rec)

    def _dw(self, name, w, b, src, out, act):
        assert len(src.segs) == 1 and src.segs[0].mode == lib.SRC_DIRECT
        self._ops.append(dict(kind=lib.OP_DWCONV, name=name, act=act, H=src.H, W=src.W, Cin=src.C, Cout=src.C, ksize=w.shape[-1],
                              segs=src.segs, out=out, out_coff=0, w=self._wput(pack.pack_dw(w, self.dtype)),
                              b=self._wput(b.float().cpu()),
                              # operands of the matrix-core variant (aux[0]) and of the pixel-pair variant (aux[1]: csrc/dwconv_p2.hip)
                              aux=[self._wput(pack.pack_dw_toeplitz(w)), self._wput(pack.pack_dw_pairs(w)) if src.C % 8 == 0 else None] if self.dtype == lib.F16 else []))

    # ---------------------------------------------------------------- graph walk
    def _build(self, model):
        B = self.B
        y = []                                   # TV (or list of head tuples) per node
        self.head_bufs = []
        self.fuse_head = self.fuse_head and not self.lanes                     # per level: head widths 64 / 128 / 192 take the fused tail
        n_side = n_head = 0
        twins_done = {}
        for node, m in zip(model.nodes, model.backbone):
            if node.i > 0:                            # tag the ops of the previous node (lane = HIP stream of the engine)
                self._tag(tag_from, tag_node, tag_lane, tag_after)
            tag_from, tag_node, tag_lane, tag_after = len(self._ops), node.i, 0, None
            p = "backbone.%d" % node.i
            srcs = [y[j] for j in node.sources()] if node.i > 0 else None
            x = srcs[0] if srcs else None
            if self.lanes >= 2 and node.kind == "cw" and node.sources()[0] != node.i - 1:
                tag_lane = 1 + (n_side % 2)           # a down-sampling conv of an OLDER map: independent of the chain in progress
                n_side += 1
            if self.lanes and node.kind == "head":
                tag_lane = (3, 5, 0)[n_head % 3]      # P3 / P4 heads overlap the rest of the neck; their reg branches take lane + 1 (P5: 7)
                tag_after = node.sources()[0]         # ... and are launched right after the node that feeds them
                n_head += 1
            if node.kind == "repvgg" and node.i == 1 and self._stem2 is not None:
                # backbone.0 + backbone.1 in one launch (csrc/stem2.hip): the half-resolution tensor between them stays in LDS
                w0, b0, c0 = self._stem2
                w, b = m.fused()
                nxt = model.nodes[2] if len(model.nodes) > 2 else None
                m2 = model.backbone[2] if nxt is not None else None
                if (nxt is not None and nxt.kind == "rephdw" and list(nxt.sources()) == [1] and not any(1 in n_.sources() for n_ in model.nodes[3:])
                        and m2.conv1.fused()[0].shape[0] == node.cout and self.fuse_stem >= 2):
                    # ... and the 1x1 that opens backbone.2 (RepHDW.conv1) rides along: node 1's tensor is never written either.
                    # Emitted when node 2 allocates its concat buffer.
                    self._stem3 = (w0, b0, c0, w, b, node.cout)
                    y.append(TV([], self.Hin // 4, self.Win // 4))
                    continue
                out = self._alloc(self.Hin // 4, self.Win // 4, node.cout)
                self._ops.append(dict(kind=lib.OP_STEM2, name="backbone.0+1", act=lib.ACT_RELU, H=out.H, W=out.W, Hin=self.Hin, Win=self.Win,
                                      Cin=3, Cout=node.cout, ksize=c0, segs=[], out=out, out_coff=0,
                                      w=self._wput(pack.pack_stem2(w0, b0, w, b)), b=0))
                y.append(TV([Seg(out, node.cout)], out.H, out.W))
            elif node.kind == "repvgg":
                w, b = m.fused()
                if node.i == 0 and self.fuse_stem and len(model.nodes) > 1 and model.nodes[1].kind == "repvgg" and list(model.nodes[1].sources()) == [0] \
                        and (node.cout, model.nodes[1].cout) in ((24, 48), (32, 64), (48, 96)) and not any(0 in n_.sources() for n_ in model.nodes[2:]):
                    self._stem2 = (w, b, node.cout)               # emitted together with node 1
                    y.append(None)
                    continue
                if node.i == 0:                   # stem: reads the caller's NCHW image
                    H, W = self.Hin // 2, self.Win // 2
                    out = self._alloc(H, W, node.cout)
                    self._ops.append(dict(kind=lib.OP_STEM, name=p, act=lib.ACT_RELU, H=H, W=W, Hin=self.Hin, Win=self.Win,
                                          Cin=3, Cout=node.cout, segs=[], out=out, out_coff=0,
                                          w=self._wput(pack.pack_stem(w)), b=self._wput(b.float().cpu())))
                else:
                    out = self._alloc(x.H // 2, x.W // 2, node.cout)
                    self._conv3x3s2(p, w, b, x, out, 0, lib.ACT_RELU)
                y.append(TV([Seg(out, node.cout)], out.H, out.W))
            elif node.kind == "rephdw":
                c_, depth = m.c_, len(m.m)
                # slot j of the concatenation (cv1's two halves, then one per block: common.py:938-946).  One interleaved buffer [.., (depth + 2) c_] —
                # or, behind the fused stem (the only producer that can write its two halves to two places), one DENSE tensor per slot: a block reads
                # and writes c_-channel slices, and out of an interleaved buffer whose slices are not whole 128-byte lines it fetches every line of the
                # buffer for a third of its bytes (bottleneck<3,1,2> on 160 x 160 x 72: 121 MB fetched for 39 MB, +13 MB of partial-line writes)
                split = (node.i == 2 and self._stem3 is not None and self.split_cat and depth + 2 <= 4 and c_ % 8 == 0
                         and (c_ * self.es) % 128 != 0 and m.conv1.fused()[0].shape[0] == 2 * c_)
                # the block's closing conv2(cat(..)) inside the launch of its LAST bottleneck: that bottleneck's output slot is then never written (nor allocated)
                kl = m.m[-1].conv2.dwconv.kernel_size
                tail = (depth <= self.fuse_tail and self._fuse_mode("%s.m.%d" % (p, depth - 1), kl, c_) == 1 and 2 <= depth + 1 <= 3
                        and m.conv2.fused()[0].shape[1] == (depth + 2) * c_
                        and lib.load().maf_bottleneck_tail_supported(kl, c_, depth + 1, node.cout) == 1)
                nslot = depth + 1 if tail else depth + 2
                if split:
                    slot = [(self._alloc(x.H, x.W, c_), 0) for _ in range(nslot)]
                else:
                    cat = self._alloc(x.H, x.W, c_ * nslot)
                    slot = [(cat, j * c_) for j in range(nslot)]
                if node.i == 2 and self._stem3 is not None:
                    w0, b0, c0, w1_, b1_, c1 = self._stem3
                    w3, b3 = m.conv1.fused()
                    self._ops.append(dict(kind=lib.OP_STEM2, name="backbone.0+1+2.conv1", act=lib.ACT_RELU, H=x.H, W=x.W, Hin=self.Hin, Win=self.Win,
                                          Cin=3, Cout=c1, ksize=c0, c3=w3.shape[0], segs=[], out=slot[0][0], out_coff=0,
                                          w=self._wput(pack.pack_stem2(w0, b0, w1_, b1_, w3, b3)), b=0))
                    if split:
                        self._ops[-1]["out2"] = slot[1][0]
                else:
                    self._conv1x1(p + ".conv1", *m.conv1.fused(), x, cat, 0, lib.ACT_SILU)
                for d, blk in enumerate(m.m):
                    mid = blk.conv1.conv.out_channels
                    q = "%s.m.%d" % (p, d)
                    mode = self._fuse_mode(q, blk.conv2.dwconv.kernel_size, c_)
                    (ib, ic) = slot[d + 1]
                    if tail and d == depth - 1:
                        # the whole DepthBottleneckUni AND the block's closing 1x1 in one launch (op.nc = the block's output channels)
                        rec, b2p, nmb, ct2 = pack.pack_bottleneck(*blk.conv1.fused(), *blk.conv2.fused(), *blk.one_conv.fused())
                        w3, b3 = m.conv2.fused()
                        out = self._alloc(x.H, x.W, node.cout)
                        self._ops.append(dict(kind=lib.OP_BOTTLENECK, name=q + "+conv2", act=lib.ACT_SILU, H=x.H, W=x.W, Cin=c_, Cout=c_, ksize=kl, mid=mid,
                                              segs=[Seg(ib, c_, ic)] + [Seg(slot[j][0], c_, slot[j][1]) for j in range(depth)], out=out, out_coff=0, pt=16, ct=16, tk=nmb,
                                              tail_c3=node.cout, w=self._wput(rec), b=self._wput(b2p), aux=[self._wput(pack.pack_bottleneck_tail(w3, b3, c_, depth + 1))]))
                        continue
                    (ob, oc) = slot[d + 2]
                    if mode == 2:
                        # conv1 + depth-wise in one launch (the 3c-wide T1 stays in LDS), then the plain 1x1
                        rec, nmb = pack.pack_conv1dw(*blk.conv1.fused(), *blk.conv2.fused())
                        t2 = self._alloc(x.H, x.W, mid)
                        self._ops.append(dict(kind=lib.OP_CONV1DW, name=q + ".conv1dw", act=lib.ACT_SILU, H=x.H, W=x.W, Cin=c_, Cout=mid,
                                              ksize=blk.conv2.dwconv.kernel_size, segs=[Seg(ib, c_, ic)], out=t2, out_coff=0, w=self._wput(rec),
                                              b=self._wput(torch.zeros(8))))
                        self._conv1x1(q + ".one_conv", *blk.one_conv.fused(), TV([Seg(t2, mid)], x.H, x.W), ob, oc, lib.ACT_SILU)
                        continue
                    if mode == 1:
                        # the whole DepthBottleneckUni in one launch; its 3c-channel intermediates stay in LDS
                        rec, b2p, nmb, ct2 = pack.pack_bottleneck(*blk.conv1.fused(), *blk.conv2.fused(), *blk.one_conv.fused())
                        k = blk.conv2.dwconv.kernel_size
                        th, tw = 16, 16                                   # fixed by the MFMA shapes of csrc/bottleneck.hip
                        self._ops.append(dict(kind=lib.OP_BOTTLENECK, name=q, act=lib.ACT_SILU, H=x.H, W=x.W, Cin=c_, Cout=c_, ksize=k, mid=mid,
                                              segs=[Seg(ib, c_, ic)], out=ob, out_coff=oc, pt=th, ct=tw, tk=nmb,
                                              w=self._wput(rec), b=self._wput(b2p), aux=[]))
                        continue
                    t1, t2 = self._alloc(x.H, x.W, mid), self._alloc(x.H, x.W, mid)
                    self._conv1x1(q + ".conv1", *blk.conv1.fused(), TV([Seg(ib, c_, ic)], x.H, x.W), t1, 0, lib.ACT_SILU)
                    self._dw(q + ".conv2", *blk.conv2.fused(), TV([Seg(t1, mid)], x.H, x.W), t2, lib.ACT_SILU)
                    self._conv1x1(q + ".one_conv", *blk.one_conv.fused(), TV([Seg(t2, mid)], x.H, x.W), ob, oc, lib.ACT_SILU)
                if not tail:
                    out = self._alloc(x.H, x.W, node.cout)
                    cat_tv = TV([Seg(b_, c_, 0) for b_, _ in slot], x.H, x.W) if split else TV([Seg(cat, c_ * (depth + 2))], x.H, x.W)
                    self._conv1x1(p + ".conv2", *m.conv2.fused(), cat_tv, out, 0, lib.ACT_SILU)
                y.append(TV([Seg(out, node.cout)], x.H, x.W))
            elif node.kind == "mprep":
                assert len(x.segs) == 1 and x.segs[0].mode == lib.SRC_DIRECT
                c_ = node.cout // 2
                out = self._alloc(x.H // 2, x.W // 2, node.cout)
                s0 = x.segs[0]
                w1_, b1_ = m.conv1.fused()
                if self.fuse_mprep and self.dtype == lib.F16 and (x.C, c_, w1_.shape[0]) in ((48, 48, 48), (64, 64, 64)) and x.H % 2 == 0 and x.W % 2 == 0 \
                        and self.B * (x.H // 2) * (x.W // 2) >= 65536:
                    # (big maps only — at bs 1 its 100 workgroups each pay the 42 KB weight prologue: n forward 0.702 -> 0.724 ms)
                    # both branches in ONE launch (csrc/conv3s2_lds.hip with nc): the 2 x 2 windows of the pooled branch lie inside the patch the
                    # 3x3 stride-2 conv stages in LDS anyway, so the input (78.6 MB at n / bs 32) is read once instead of twice
                    w2_, b2_ = m.conv2.fused()
                    self._ops.append(dict(kind=lib.OP_CONV3X3S2, name=p + ".conv1+conv2", act=lib.ACT_RELU, H=x.H // 2, W=x.W // 2, Hin=x.H, Win=x.W, Cin=x.C, Cout=c_,
                                          raw=(w2_.detach().float().cpu(), b2_.detach().float().cpu(), None), pool1=(w1_.detach().float().cpu(), b1_.detach().float().cpu()),
                                          segs=x.segs, out=out, out_coff=c_, out_f32=0, pt=4, ct=4, w=self._wput(pack.pack_mprep_lds(w2_, b2_, w1_, b1_)), b=0))
                    y.append(TV([Seg(out, node.cout)], out.H, out.W))
                    continue
                if self.fuse_mprep and self.dtype == lib.F16 and (x.C, c_, w1_.shape[0]) == (96, 96, 96) and x.H % 2 == 0 and x.W % 2 == 0 \
                        and self.B * (x.H // 2) * (x.W // 2) >= cfg.mprep_wreg_min:
                    # (backbone.5 of n at bs 32, 51 200 pixels: 24.9 + 16.7 -> 38.8 us only — the pooled branch costs this kernel its read-ahead depth — so big maps only)
                    # ... and on the register-resident 3x3 kernel (csrc/conv3s2_wreg.hip with nc): the pooled operand is the maximum of four fragments the conv reads anyway
                    w2_, b2_ = m.conv2.fused()
                    self._ops.append(dict(kind=lib.OP_CONV3X3S2, name=p + ".conv1+conv2", act=lib.ACT_RELU, H=x.H // 2, W=x.W // 2, Hin=x.H, Win=x.W, Cin=x.C, Cout=c_,
                                          raw=(w2_.detach().float().cpu(), b2_.detach().float().cpu(), None), pool1=(w1_.detach().float().cpu(), b1_.detach().float().cpu()), pool1_tk=7,
                                          segs=x.segs, out=out, out_coff=c_, out_f32=0, pt=2, ct=8, w=self._wput(pack.pack_mprep_wreg(w2_, b2_, w1_, b1_)), b=0))
                    y.append(TV([Seg(out, node.cout)], out.H, out.W))
                    continue
                pooled = TV([Seg(s0.buf, s0.C, s0.coff, lib.SRC_POOL2)], x.H // 2, x.W // 2)
                self._conv1x1(p + ".conv1", *m.conv1.fused(), pooled, out, 0, lib.ACT_SILU)
                self._conv3x3s2(p + ".conv2", *m.conv2.fused(), x, out, c_, lib.ACT_RELU)
                y.append(TV([Seg(out, node.cout)], out.H, out.W))
            elif node.kind == "sppf":
                c_ = m.cv1.conv.out_channels
                cat = self._alloc(x.H, x.W, 4 * c_)
                self._conv1x1(p + ".cv1", *m.cv1.fused(), x, cat, 0, lib.ACT_SILU)
                self._ops.append(dict(kind=lib.OP_SPPF_POOL, name=p + ".m", act=0, H=x.H, W=x.W, Cin=c_, Cout=3 * c_,
                                      segs=[Seg(cat, c_, 0)], out=cat, out_coff=c_))
                out = self._alloc(x.H, x.W, node.cout)
                self._conv1x1(p + ".cv2", *m.cv2.fused(), TV([Seg(cat, 4 * c_)], x.H, x.W), out, 0, lib.ACT_SILU)
                y.append(TV([Seg(out, node.cout)], x.H, x.W))
            elif node.kind == "cw":
                if node.i in twins_done:                  # emitted together with the previous node
                    y.append(twins_done.pop(node.i))
                    continue
                out = self._alloc(x.H // 2, x.W // 2, node.cout)
                # the two side convs of a MAFPN level (backbone.23 / .24, .27 / .28) are independent and equal in shape: ONE launch
                twin = None
                nxt = model.nodes[node.i + 1] if node.i + 1 < len(model.nodes) else None
                if getattr(model, "twin_convs", True) and not self.lanes and nxt is not None and nxt.kind == "cw" and (nxt.cin, nxt.cout) == (node.cin, node.cout) \
                        and node.i not in nxt.sources() and all(j < node.i for j in nxt.sources()):
                    x2 = y[nxt.sources()[0]]
                    if (x2.H, x2.W) == (x.H, x.W) and len(x2.segs) == 1 and len(x.segs) == 1 and x2.segs[0].mode == x.segs[0].mode == lib.SRC_DIRECT \
                            and (x2.segs[0].buf.stride, x2.segs[0].coff) == (x.segs[0].buf.stride, x.segs[0].coff):
                        out2 = self._alloc(x.H // 2, x.W // 2, nxt.cout)
                        twin = ("backbone.%d.block" % nxt.i, *model.backbone[nxt.i].block.fused(), x2, out2)
                        twins_done[nxt.i] = TV([Seg(out2, nxt.cout)], out2.H, out2.W)
                self._conv3x3s2(p + ".block", *m.block.fused(), x, out, 0, lib.ACT_SILU, twin=twin)
                y.append(TV([Seg(out, node.cout)], out.H, out.W))
            elif node.kind == "concat":
                H, W = srcs[0].H, srcs[0].W
                segs = []
                for s in srcs:
                    assert (s.H, s.W) == (H, W), "concat of different grids at node %d" % node.i
                    segs += s.segs
                y.append(TV(segs, H, W))
            elif node.kind == "up":
                assert all(s.mode == lib.SRC_DIRECT for s in x.segs), "upsample of a non-materialised tensor"
                y.append(TV([Seg(s.buf, s.C, s.coff, lib.SRC_UP2) for s in x.segs], x.H * 2, x.W * 2))
            elif node.kind == "head":
                c = node.cout
                t = self._alloc(x.H, x.W, c)
                self._conv1x1(p + ".stem", *m.stem.fused(), x, t, 0, lib.ACT_SILU)
                tv = TV([Seg(t, c)], x.H, x.W)
                # 64 / 128 / 192: weights LDS-resident; 256: weight chunks streamed through LDS once per 16-pixel unit — pays on small levels
                # (the bs = 1 latency path, P5 of s); 384 (P4 / P5 of m) keeps the convs + decode kernel (its instantiation spills)
                if self.fuse_head and (c in (64, 128, 192) or (c == 256 and self.B * x.H * x.W <= cfg.head_tail_256_max)):
                    # cls_conv and reg_conv read the same tensor: ONE depth-wise launch with two filters per input channel
                    (wc, bc), (wr, brg) = m.cls_conv.fused(), m.reg_conv.fused()
                    assert wc.shape == wr.shape
                    u = self._alloc(x.H, x.W, 2 * c)
                    self._ops.append(dict(kind=lib.OP_DWCONV, name=p + ".cls_reg_conv", act=lib.ACT_NONE, H=x.H, W=x.W, Cin=c, Cout=2 * c, ksize=wc.shape[-1],
                                          segs=tv.segs, out=u, out_coff=0, w=self._wput(pack.pack_dw(torch.cat([wc, wr], 0), self.dtype)),
                                          b=self._wput(torch.cat([bc, brg], 0).float().cpu()),
                                          aux=[None, self._wput(pack.pack_dw_pairs(torch.cat([wc, wr], 0)))] if self.dtype == lib.F16 and c % 8 == 0 else []))
                    us = [(u, 0), (u, c)]
                    recs = [pack.pack_head_tail(*getattr(m, br + "_conv_s").fused(), pr.weight.detach(), pr.bias.detach())
                            for br, pr in (("cls", m.cls_pred), ("reg", m.reg_pred))]
                    self._ops.append(dict(kind=lib.OP_HEADTAIL, name=p + ".tail", act=0, H=x.H, W=x.W, Cin=c, Cout=5 + self.nc,
                                          segs=[Seg(us[0][0], c, us[0][1]), Seg(us[1][0], c, us[1][1])], out=None, out_coff=0, w=self._wput(recs[0]), b=0,
                                          aux=[self._wput(recs[1])], level=len(self.head_bufs)))
                    self.head_bufs.append((t, None, None))
                    y.append(None)
                    continue
                outs = []
                u2 = None
                if self.dtype == lib.F16 and c % 8 == 0:
                    # cls_conv and reg_conv read the same tensor here too: ONE depth-wise launch with two filters per input channel (and, as the
                    # only reader of the stem's output, a candidate for the pixel-pair variant)
                    (wc, bc), (wr, brg) = m.cls_conv.fused(), m.reg_conv.fused()
                    if wc.shape == wr.shape:
                        u2 = self._alloc(x.H, x.W, 2 * c)
                        self._ops.append(dict(kind=lib.OP_DWCONV, name=p + ".cls_reg_conv", act=lib.ACT_NONE, H=x.H, W=x.W, Cin=c, Cout=2 * c, ksize=wc.shape[-1],
                                              segs=tv.segs, out=u2, out_coff=0, w=self._wput(pack.pack_dw(torch.cat([wc, wr], 0), self.dtype)),
                                              b=self._wput(torch.cat([bc, brg], 0).float().cpu()),
                                              aux=[None, self._wput(pack.pack_dw_pairs(torch.cat([wc, wr], 0)))]))
                for bi, (br, pred, act, cpred) in enumerate((("cls", m.cls_pred, lib.ACT_SIGMOID, self.nc), ("reg", m.reg_pred, lib.ACT_NONE, 4 * (self.reg_max + 1)))):
                    u, v = (None if u2 is not None else self._alloc(x.H, x.W, c)), self._alloc(x.H, x.W, c)
                    # any class count (the reference takes any nc): the conv kernels store 4 channels at a time, so the pred conv is
                    # padded with zero filters to a multiple of 4 and the decode kernel reads the rows with that stride
                    cpad = -(-cpred // 4) * 4
                    o = self._alloc(x.H, x.W, cpad, 4)                                    # fp32 [B,HW,cpad]
                    o.C = cpred
                    pw, pb = pred.weight.detach(), pred.bias.detach()
                    if cpad != cpred:
                        pw = torch.cat([pw, pw.new_zeros(cpad - cpred, *pw.shape[1:])], 0)
                        pb = torch.cat([pb, pb.new_zeros(cpad - cpred)], 0)
                    if u2 is None:
                        self._dw("%s.%s_conv" % (p, br), *getattr(m, br + "_conv").fused(), tv, u, lib.ACT_NONE)
                    self._conv1x1("%s.%s_conv_s" % (p, br), *getattr(m, br + "_conv_s").fused(),
                                  TV([Seg(u, c)] if u2 is None else [Seg(u2, c, bi * c)], x.H, x.W), v, 0, lib.ACT_SILU)
                    self._conv1x1("%s.%s_pred" % (p, br), pw, pb, TV([Seg(v, c)], x.H, x.W), o, 0, act, out_f32=True)
                    outs.append(o)
                self.head_bufs.append((t, outs[0], outs[1]))
                y.append(None)
            elif node.kind == "out":
                y.append(None)
            else:
                raise NotImplementedError(node.kind)
        self._tag(tag_from, tag_node, tag_lane, tag_after)
        assert len(self.head_bufs) == 3, "MAF-YOLO has three detection levels"
        self.A = sum(t.H * t.W for t, _, _ in self.head_bufs)
        fused = [c is None for _, c, _ in self.head_bufs]
        if not all(fused):                           # the decode kernel skips the levels a fused tail has written
            self._ops.append(dict(kind=lib.OP_DECODE, name="detect", act=0, H=0, W=0, Cin=0, Cout=0, segs=[], out=None, out_coff=0))

    # ---------------------------------------------------------------- lanes and cross-lane dependencies
    def _tag(self, first, node_i, lane, after):
        for r in self._ops[first:]:
            r["node"], r["after"] = node_i, after
            r["lane"] = lane
            if lane is not None and ".reg_" in r["name"] and self.lanes:          # head: the reg branch runs beside the cls branch
                r["lane"] = 7 if lane == 0 else lane + 1

    def _schedule(self):
        """Launch order + event waits.  The op list stays a topological order (single-stream execution of it is always valid:
        run_timed does that); heads move up to just behind the node that feeds them so their lanes start early.  For every op
        the ops of OTHER lanes whose output slices it reads become `wait` entries (latest one per lane, transitively pruned)."""
        ops = self._ops
        if self.lanes:
            rest = [r for r in ops if r.get("after") is None]
            for src_node in sorted({r["after"] for r in ops if r.get("after") is not None}):
                grp = [r for r in ops if r.get("after") == src_node]          # one head, in its emission order
                anchor = max(i for i, q in enumerate(rest) if q.get("node") == src_node)
                rest[anchor + 1:anchor + 1] = grp
            ops = self._ops = rest
        writes = {}                                                # id(buf) -> [(lo, hi, op index)]
        seen = {}                                                  # lane -> {other lane: latest op index already waited for}
        last_on_lane = {}
        for i, r in enumerate(ops):
            lane = r.get("lane", 0) if self.lanes else 0
            r["lane"] = lane
            deps = set()
            if r["kind"] == lib.OP_DECODE:
                deps = set(last_on_lane.values())
            for s_ in r["segs"]:
                for lo, hi, j in writes.get(id(s_.buf), []):
                    if lo < s_.coff + s_.C and s_.coff < hi:
                        deps.add(j)
            latest = {}
            for j in deps:
                lj = ops[j]["lane"]
                if lj != lane and j > seen.setdefault(lane, {}).get(lj, -1):
                    latest[lj] = max(latest.get(lj, -1), j)
            # what the awaited ops had themselves waited for is ordered before them: inherit it
            for lj, j in latest.items():
                seen[lane][lj] = j
                for lk, jk in seen.get(lj, {}).items():
                    if lk != lane:
                        seen[lane][lk] = max(seen[lane].get(lk, -1), min(jk, j))
            r["wait"] = sorted(latest.values())
            assert len(r["wait"]) <= 8
            if r["out"] is not None:
                writes.setdefault(id(r["out"]), []).append((r["out_coff"], r["out_coff"] + r["Cout"], i))
            if "twin" in r:
                writes.setdefault(id(r["twin"]["out"]), []).append((0, r["Cout"], i))
            if "out2" in r:
                writes.setdefault(id(r["out2"]), []).append((0, r["Cout"], i))
            if "pool1" in r:                                  # the pooled branch of a one-launch MPRep writes channels 0 .. C1 of the same pixels
                writes.setdefault(id(r["out"]), []).append((0, r["pool1"][0].shape[0], i))
            last_on_lane[lane] = i

    # ---------------------------------------------------------------- materialise
    def _finalize(self):
        self._schedule()
        dev = self.device
        self.arena = torch.empty(self._arena_size + 256, dtype=torch.uint8, device=dev)
        wcpu = torch.zeros(self._wsize + 256, dtype=torch.uint8)
        for off, t in self._wblobs:
            wcpu[off:off + t.numel() * t.element_size()] = t.reshape(-1).view(torch.uint8)
        self.weights = wcpu.to(dev)
        self._wblobs = None
        abase, wbase = self.arena.data_ptr(), self.weights.data_ptr()
        abase += (-abase) % 256
        self._abase = abase
        ops = (lib.MafOp * len(self._ops))()
        for o, r in zip(ops, self._ops):
            o.kind, o.dtype, o.in_dtype, o.act = r["kind"], self.dtype, self.in_dtype, r["act"]
            o.B, o.H, o.W = self.B, r["H"], r["W"]
            o.Hin, o.Win = r.get("Hin", 0), r.get("Win", 0)
            o.Cin, o.Cout, o.ksize = r["Cin"], r["Cout"], r.get("ksize", 0)
            o.nsrc = len(r["segs"])
            for i, s in enumerate(r["segs"]):
                o.src[i].ptr = abase + s.buf.off
                o.src[i].C, o.src[i].stride, o.src[i].coff, o.src[i].mode = s.C, s.buf.stride, s.coff, s.mode
            if r["out"] is not None:
                o.out = abase + r["out"].off
                o.out_stride = r["out"].stride
            o.out_coff = r["out_coff"]
            o.out_f32 = r.get("out_f32", 0)
            o.tile_p, o.tile_c, o.tile_k = r.get("pt", 0), r.get("ct", 0), (1 if r["kind"] in (lib.OP_CONV1X1, lib.OP_CONV3X3S2) else 0)
            o.ksize = r.get("ksize", 0)
            if "w" in r:
                o.w = wbase + r["w"]
                o.bias = wbase + r["b"]
            if r["kind"] == lib.OP_BOTTLENECK:
                o.tile_k = r["tk"]
                o.nc = r.get("tail_c3", 0)
            if "pool1" in r:                                  # one-launch MPRep: only the LDS-resident 3x3 kernel has the pooled branch
                o.tile_k, o.nc, o.reg_stride = r.get("pool1_tk", 6), r["pool1"][0].shape[0], 0
            for k_, off in enumerate(r.get("aux", [])):
                if off is not None:
                    o.aux[k_] = wbase + off
            if "twin" in r:                                   # second conv of a twin launch: {src, w, bias, out}
                tw = r["twin"]
                o.aux[0], o.aux[1], o.aux[2], o.aux[3] = abase + tw["seg"].buf.off, wbase + tw["w"], wbase + tw["b"], abase + tw["out"].off
            o.lane, o.n_wait = r["lane"], len(r["wait"])
            for k_, j in enumerate(r["wait"]):
                o.wait[k_] = j
            if r["kind"] in (lib.OP_STEM, lib.OP_STEM2):
                o.nsrc = 1
                o.src[0].ptr = 0            # supplied per call
                o.src[0].C = 3
            if r["kind"] == lib.OP_STEM2:
                o.nc = r.get("c3", 0)
                if "out2" in r:              # the upper half of the channels as a tensor of its own
                    o.aux[0] = abase + r["out2"].off
                    o.reg_stride = r["out2"].stride
            if r["kind"] == lib.OP_HEADTAIL:
                lvl = r["level"]
                o.Hin, o.Win = sum(t.H * t.W for t, _, _ in self.head_bufs[:lvl]), self.A
                o.lvl_stride[0], o.nc, o.reg_max = self.strides[lvl], self.nc, self.reg_max
            if r["kind"] == lib.OP_DECODE:
                o.nsrc = 3
                for l, (t, cls, reg) in enumerate(self.head_bufs):
                    if cls is not None:
                        o.src[l].ptr = abase + cls.off
                        o.src[l].stride = cls.stride                      # row stride of the class scores (nc rounded up to 4)
                        o.reg[l] = abase + reg.off
                    o.lvl_h[l], o.lvl_w[l], o.lvl_stride[l] = t.H, t.W, self.strides[l]
                o.reg_stride, o.nc, o.reg_max = -(-4 * (self.reg_max + 1) // 4) * 4, self.nc, self.reg_max
        self.ops = ops
        self.op_names = [r["name"] for r in self._ops]
        h = C.c_void_p()
        lib.check(lib.load().maf_engine_create(ops, len(self._ops), C.byref(h)))
        self._engine = h

    def _pairs_producer(self, i):
        """The op that may hand op i (a depth-wise conv) its input as PIXEL PAIRS (lib.SRC_PAIRS, csrc/dwconv_p2.hip), or None: the 1x1 conv right in front
        of it, running as conv1x1_stream_lds (the variant with the pair epilogue), writing exactly the buffer op i reads — and nobody else reads it."""
        o = self.ops[i]
        if i == 0 or self.dtype != lib.F16 or o.kind != lib.OP_DWCONV or not o.aux[1] or o.W % 2 or o.Cin % 8 or o.nsrc != 1 or o.src[0].mode not in (lib.SRC_DIRECT, lib.SRC_PAIRS):
            return None
        p = self.ops[i - 1]
        if p.kind != lib.OP_CONV1X1 or p.tile_k != 5 or p.out_f32 or "twin" in self._ops[i - 1] or p.out != o.src[0].ptr or p.out_coff or o.src[0].coff \
                or p.Cout != o.Cin or p.out_stride != o.src[0].stride or p.out_stride % 4 or (p.B, p.H, p.W) != (o.B, o.H, o.W):
            return None
        for j, q in enumerate(self.ops):                                    # any other reader (or writer) of the buffer keeps it NHWC
            if j in (i, i - 1):
                continue
            if q.out == p.out or any(q.src[k].ptr == p.out for k in range(q.nsrc)) or any(q.aux[k] == p.out for k in range(4)):
                return None
        return p

    # ---------------------------------------------------------------- tile autotuning (tuner.py)
    def autotune(self, x, reps=5, verbose=False):
        """Time every tile / variant candidate of every conv and depth-wise launch on this device and keep the fastest (tuner.autotune)."""
        return _tuner.autotune(self, x, reps, verbose)

    # ---------------------------------------------------------------- execution
    def run(self, x, graph=False):
        """x: [B,3,Hin,Win] contiguous NCHW tensor on self.device. Returns pred fp32 [B, A, 5+nc]."""
        pred = torch.empty(self.B, self.A, 5 + self.nc, dtype=torch.float32, device=self.device)
        return self.run_into(x, pred, graph)

    def filter_ok(self):
        """True if this plan can run non_max_suppression's candidate filter inside its head tails (maf_engine_run_filtered): every level ends
        in a fused MAF_OP_HEADTAIL with H * W a multiple of 16."""
        tails = [o for o in self.ops if o.kind == lib.OP_HEADTAIL]
        return len(tails) == 3 and not any(o.kind == lib.OP_DECODE for o in self.ops) and all((o.H * o.W) % 16 == 0 for o in tails)

    def run_into(self, x, pred, graph=False, cand=None):
        """`cand` = (workspace tensor, conf_thres): also fill the NMS candidate lists of that workspace (see Model.nms_filter)."""
        cur = torch.cuda.current_stream(self.device)
        if not graph:
            if cand is not None:
                lib.check(lib.load().maf_engine_run_filtered(self._engine, x.data_ptr(), pred.data_ptr(), cur.cuda_stream, cand[0].data_ptr(), float(cand[1])))
            else:
                lib.check(lib.load().maf_engine_run(self._engine, x.data_ptr(), pred.data_ptr(), cur.cuda_stream))
            return pred
        # hipGraph capture is not permitted on the legacy default stream: replay on a private stream ordered
        # after / before the caller's current stream.  Pointers are frozen at capture (same x / pred buffers).
        gs = getattr(self, "_gstream", None)
        if gs is None:
            gs = self._gstream = torch.cuda.Stream(self.device)
        gs.wait_stream(cur)
        lib.check(lib.load().maf_engine_run_graph(self._engine, x.data_ptr(), pred.data_ptr(), gs.cuda_stream))
        cur.wait_stream(gs)
        return pred

    def run_timed(self, x, pred):
        """One forward with HIP events around every op -> list of per-op milliseconds (bench.py roofline)."""
        ms = (C.c_float * len(self.ops))()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        lib.check(lib.load().maf_engine_run_timed(self._engine, x.data_ptr(), pred.data_ptr(), stream, ms))
        return list(ms)

    # ---------------------------------------------------------------- what the plan says about itself (plan_report.py)
    def kernel_name(self, idx):
        return _report.kernel_name(self, idx)

    def algorithmic_bytes(self, idx):
        return _report.algorithmic_bytes(self, idx)

    def flops(self, idx):
        return _report.flops(self, idx)

    def launch_op(self, idx, image_ptr=None, pred_ptr=None):
        """Launch a single op of the plan (profiling / per-kernel timing)."""
        op = self.ops[idx]
        if op.kind in (lib.OP_STEM, lib.OP_STEM2) and image_ptr is not None:
            op.src[0].ptr = image_ptr
        if op.kind in (lib.OP_DECODE, lib.OP_HEADTAIL) and pred_ptr is not None:
            op.out = pred_ptr
        stream = torch.cuda.current_stream(self.device).cuda_stream
        lib.check(lib.load().maf_op_launch(C.byref(op), stream))

    def view(self, buf):
        """Zero-copy torch view [B,H,W,stride] of an arena buffer (valid until the next forward)."""
        dt = torch.float32 if buf.esize == 4 else torch.float16
        start = buf.off + (self._abase - self.arena.data_ptr())
        n = self.B * buf.H * buf.W * buf.stride * buf.esize
        return self.arena[start:start + n].view(dt).view(self.B, buf.H, buf.W, buf.stride)

    def featmaps(self):
        """[(stem, cls, reg)] x 3 as NCHW views, the second element of the reference's Model.forward return.  Plans with the fused
        head tail never materialise cls / reg (None there): Model.forward(val_loss=True) builds its plan with fuse_head off."""
        return [tuple(None if b is None else self.view(b)[..., :b.C].permute(0, 3, 1, 2) for b in hb) for hb in self.head_bufs]

    def __del__(self):
        try:
            if getattr(self, "_engine", None):
                lib.load().maf_engine_destroy(self._engine)
        except Exception:
            pass
