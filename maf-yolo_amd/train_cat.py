"""Training-form data movement on the HIP kernels: copy-free concats (CatBuffer / join / fork), gradient sums of multi-reader tensors (fanout), Detect's train-branch join
(yolov6/models/yolo.py:333-354), max-pooling (MP, SPPF.m) and nearest 2x up-sampling, each as an autograd Function a step tape can record.

One of the four family files train_ops.py was cut into in round 6 (train_conv / train_dw / train_bn / train_cat).  `T` is train_ops itself: every module-level switch, cache and
helper lives THERE (tests, tools and tape.py read and set them as `train_ops.<name>`), and every reference from here goes through `T.<name>` at call time — so a switch flipped or
an entry point replaced on train_ops (bench.py --torch-convs) reaches this code exactly as it did when all of it was one file.  train_ops re-exports everything defined here;
import train_ops (or the package), not this file."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import lib, pack
from . import train_ops as T


def cat_free_ok(x, bn):
    """The copy-free concat needs the HIP BatchNorm path for the producers."""
    return T.cat_free and x.is_cuda and bn.training and not T.framework_ops and x.dtype in T._DT


class Like:
    """shape / dtype / device of a tensor that does not exist yet (what CatBuffer needs of its `like`)."""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = tuple(shape), dtype, device


class CatBuffer:
    """One NHWC tensor for a channel concat whose producers store into their slots.  `like`: a tensor with the concat's batch, spatial size, dtype, device."""

    def __init__(self, like, widths):
        B, _, H, W = like.shape
        self.offs = [0]
        for w in widths:
            self.offs.append(self.offs[-1] + w)
        self.buf = T._empty((B, self.offs[-1], H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)

    def slot(self, i, n=1):
        """Channels of slots i .. i + n - 1 as a tensor of its own on the buffer's storage — NOT a view of `buf` for autograd: a slot becomes the output of its
        producer's autograd node, and a view whose base is written through another view later (join's copies) is refused there."""
        b = self.buf
        t = T._empty(0, dtype=b.dtype, device=b.device)
        t.set_(b.untyped_storage(), b.storage_offset() + self.offs[i], (b.shape[0], self.offs[i + n] - self.offs[i], b.shape[2], b.shape[3]), b.stride())
        return t


@T._laned
class _Join(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cb, *parts):
        ctx.offs = [0]
        for p_ in parts:
            ctx.offs.append(ctx.offs[-1] + p_.shape[1])
        if ctx.offs[-1] != cb.buf.shape[1]:
            raise lib.MafError("join: the parts' channels must add up to the buffer's")
        es = cb.buf.element_size()
        for i, (p_, o) in enumerate(zip(parts, ctx.offs)):
            if p_.data_ptr() != cb.buf.data_ptr() + o * es or p_.stride() != cb.buf.stride():     # not stored there by its producer (e.g. a map a second concat lists): one strided copy
                # (the channel range comes from the parts' own widths — ctx.offs, what backward slices by — not from the buffer's slot table: a part may span several slots)
                b_ = cb.buf
                dst = T._empty(0, dtype=b_.dtype, device=b_.device)
                dst.set_(b_.untyped_storage(), b_.storage_offset() + o, (b_.shape[0], p_.shape[1], b_.shape[2], b_.shape[3]), b_.stride())
                mult = 8 if b_.dtype == torch.float16 else 4
                if b_.is_cuda and not T.framework_ops and b_.dtype in T._DT and p_.shape[1] % mult == 0 and o % mult == 0 and T.nhwc(p_)[0] is p_:
                    T.nhwc_sum([p_], dst)
                else:
                    T._glue()
                    dst.copy_(p_)
                T.stats["cat_copied_parts"] = T.stats.get("cat_copied_parts", 0) + 1
        T.stats["cat_free"] = T.stats.get("cat_free", 0) + 1
        return cb.buf

    @staticmethod
    def backward(ctx, d):
        return (None,) + tuple(d[:, a:b] for a, b in zip(ctx.offs[:-1], ctx.offs[1:]))


def join(cb, parts):
    """The concat of `parts` along the channels in CatBuffer `cb`: parts their producer stored into their slot (`bn_act(out=cb.slot(i))`) cost nothing, any
    other part is copied into its slot."""
    if any(p_.dtype != cb.buf.dtype or p_.shape[0] != cb.buf.shape[0] or p_.shape[2:] != cb.buf.shape[2:] or p_.device != cb.buf.device for p_ in parts):
        if cb.buf.is_cuda:
            T._glue()
        return torch.cat(parts, 1)                                               # (mixed dtypes promote: the framework's rule)
    return T._Join.apply(cb, *parts)


@T._laned
class _Fork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, lo):
        ctx.lo = lo
        return t.view_as(t), t[:, lo:]

    @staticmethod
    def backward(ctx, d_all, d_tail):
        if d_all is None:
            if d_tail is None:
                return None, None
            T._glue()                                                               # the zero fill is a torch kernel that would run at recording time only: a step tape refuses this step
            d_all = T._tzeros((d_tail.shape[0], ctx.lo + d_tail.shape[1]) + tuple(d_tail.shape[2:]), dtype=d_tail.dtype, device=d_tail.device).contiguous(memory_format=torch.channels_last)
        if d_tail is not None:
            tgt = d_all[:, ctx.lo:]
            mult = 8 if d_all.dtype == torch.float16 else 4
            if (d_all.is_cuda and not T.framework_ops and d_all.dtype in T._DT and d_tail.dtype == d_all.dtype and d_tail.shape[1] % mult == 0 and ctx.lo % mult == 0
                    and T.nhwc(tgt)[0] is tgt and T.nhwc(d_tail)[0] is d_tail):
                T.nhwc_sum([d_tail], tgt, accumulate=True)                          # one launch on NHWC views (csrc/train_ops.hip maf_nhwc_sum), recordable by a step tape
            else:
                T._glue()
                tgt.add_(d_tail)
        return d_all, None


class _DetectJoin(torch.autograd.Function):
    """Detect_yaml's train branch (yolov6/models/yolo.py:333-354) + the head's class sigmoid (yolov6/layers/common.py:1332): per-level NHWC (logits, box distribution)
    maps -> (cls [B,A,nc] probabilities, reg [B,A,4*(reg_max+1)]) in ONE launch (csrc/detect_join.hip), and one launch back: d logits = d cls * y * (1 - y), d reg, into
    per-level gradient maps padded to the conv kernels' 16-byte channel group (pad channels written as zeros: no memset, no F.pad in front of the weight gradient).
    A step tape records both calls."""

    @staticmethod
    def forward(ctx, nl, *ts):
        cls_l, reg_l = [T.nhwc(t) for t in ts[0::2]], [T.nhwc(t) for t in ts[1::2]]
        t0 = cls_l[0][0]
        B, nc = t0.shape[:2]
        nr = reg_l[0][0].shape[1]
        hw = [t.shape[2] * t.shape[3] for t, _ in cls_l]
        A = sum(hw)
        cls = T._empty((B, A, nc), dtype=t0.dtype, device=t0.device)
        reg = T._empty((B, A, nr), dtype=t0.dtype, device=t0.device)
        P, I = C.c_void_p * nl, C.c_int32 * nl
        hwa = I(*hw)
        with T._prof("detect_join", 2 * B * A * (nc + nr) * t0.element_size(), t0.device):
            lib.check(lib.load().maf_detect_join(P(*[t.data_ptr() for t, _ in cls_l]), I(*[s_ for _, s_ in cls_l]), P(*[t.data_ptr() for t, _ in reg_l]), I(*[s_ for _, s_ in reg_l]),
                                                 hwa, nl, B, nc, nr, T._DT[t0.dtype], cls.data_ptr(), reg.data_ptr(), T._stream(t0.device)))
        T.stats["native_detect_join"] = T.stats.get("native_detect_join", 0) + 1
        ctx.save_for_backward(cls)
        ctx.geo = (nl, B, nc, nr, hw, [tuple(t.shape[2:]) for t, _ in cls_l])
        return cls, reg

    @staticmethod
    def backward(ctx, d_cls, d_reg):
        (cls,) = ctx.saved_tensors
        nl, B, nc, nr, hw, shapes = ctx.geo
        dt, dev = cls.dtype, cls.device
        for name, d in (("cls", d_cls), ("reg", d_reg)):
            if d is not None and (d.dtype != dt or not d.is_contiguous()):
                T._glue()                                                           # (the loss kernels and a step tape's boundary hand over contiguous tensors of the head's dtype)
        d_cls = None if d_cls is None else d_cls.to(dt).contiguous()
        d_reg = None if d_reg is None else d_reg.to(dt).contiguous()
        mult = 8 if dt == torch.float16 else 4
        ncp, nrp = -(-nc // mult) * mult, -(-nr // mult) * mult
        outs, dc, dr = [], [], []
        for h, w in shapes:
            gc = T._empty((B, ncp, h, w), dtype=dt, device=dev, memory_format=torch.channels_last)
            gr = T._empty((B, nrp, h, w), dtype=dt, device=dev, memory_format=torch.channels_last)
            dc.append(gc); dr.append(gr)
            vc, vr = (gc[:, :nc] if ncp != nc else gc), (gr[:, :nr] if nrp != nr else gr)
            if ncp != nc:
                T.zero_padded[vc.data_ptr()] = ncp                                  # the channels behind the view are zeros (the kernel writes them): _wgrad / the data gradient read whole groups
            if nrp != nr:
                T.zero_padded[vr.data_ptr()] = nrp
            outs += [vc, vr]
        P, I = C.c_void_p * nl, C.c_int32 * nl
        with T._prof("detect_join_backward", B * sum(hw) * (3 * nc + 2 * nr) * cls.element_size(), dev):
            lib.check(lib.load().maf_detect_join_backward(None if d_cls is None else d_cls.data_ptr(), None if d_reg is None else d_reg.data_ptr(), cls.data_ptr(), I(*hw), nl, B, nc, nr,
                                                          T._DT[dt], P(*[t.data_ptr() for t in dc]), I(*[ncp] * nl), P(*[t.data_ptr() for t in dr]), I(*[nrp] * nl), ncp, nrp, T._stream(dev)))
        T.stats["native_detect_join"] = T.stats.get("native_detect_join", 0) + 1
        if T._keep is not None:
            T._keep.extend([d_cls, d_reg])
        return (None,) + tuple(outs)


def detect_join(heads):
    """(cls [B,A,nc] class probabilities, reg [B,A,4*(reg_max+1)]) from the per-level (stem, class LOGITS, box distribution) of the heads: Detect_yaml's train branch
    (yolov6/models/yolo.py:333-354: flatten + permute + cat) with the class sigmoid of Head_DepthUni (yolov6/layers/common.py:1332) folded in.  HIP tensors: one launch
    (csrc/detect_join.hip); CPU tensors / `framework_ops`: the reference's torch ops."""
    cls_l, reg_l = [h[1] for h in heads], [h[2] for h in heads]
    t0 = cls_l[0]
    native = (t0.is_cuda and not T.framework_ops and len(heads) <= 4 and t0.shape[1] % 4 == 0 and reg_l[0].shape[1] % 4 == 0
              and all(t.dim() == 4 and t.dtype == t0.dtype and t.dtype in T._DT for t in cls_l + reg_l))
    if not native:
        if t0.is_cuda and not T.framework_ops:
            T._glue(); T.stats["fallback"] += 1
        cls = torch.cat([torch.sigmoid(c).flatten(2).permute(0, 2, 1) for c in cls_l], 1)
        reg = torch.cat([r.flatten(2).permute(0, 2, 1) for r in reg_l], 1)
        return cls, reg
    return T._DetectJoin.apply(len(heads), *[t for pair in zip(cls_l, reg_l) for t in pair])


def nhwc_sum(srcs, dst, accumulate=False):
    """dst = [dst +] sum(srcs) on NHWC views of one shape and dtype (csrc/train_ops.hip maf_nhwc_sum: 1..4 sources, channel slices welcome)."""
    B, c, H, W = dst.shape
    n = len(srcs)
    lib.check(lib.load().maf_nhwc_sum(T._PTR4(*[t.data_ptr() for t in srcs]), T._INT4(*[t.stride()[3] for t in srcs]), n, dst.data_ptr(), dst.stride()[3],
                                      B * H * W, c, T._DT[dst.dtype], 1 if accumulate else 0, T._stream(dst.device)))
    T.stats["native_nhwc_sum"] = T.stats.get("native_nhwc_sum", 0) + 1


@T._laned
class _Fanout(torch.autograd.Function):
    """n aliases of one tensor for n consumers; backward = the sum of their gradients in ONE launch (fp32 sum, one rounding) instead of the autograd
    engine's add kernel per extra consumer — and a launch a step tape can record."""

    @staticmethod
    def forward(ctx, t, n):
        ctx.n = n
        return tuple(t.view_as(t) for _ in range(n))

    @staticmethod
    def backward(ctx, *ds):
        live = [d for d in ds if d is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        d0 = live[0]
        mult = 8 if d0.dtype == torch.float16 else 4
        if (d0.is_cuda and not T.framework_ops and d0.dtype in T._DT and d0.dim() == 4 and d0.shape[1] % mult == 0 and len(live) <= 4
                and all(d.dtype == d0.dtype and d.shape == d0.shape for d in live)):
            views = [T.nhwc(d)[0] for d in live]
            out = T._empty(d0.shape, dtype=d0.dtype, device=d0.device, memory_format=torch.channels_last)
            T.nhwc_sum(views, out)
            return out, None
        T._glue(len(live) - 1)
        out = live[0] + live[1]
        for d in live[2:]:
            out = out + d
        return out, None


def fanout(t, n):
    """[t] * n for a tensor with n consumers inside the train-form graph (a backbone map the neck reads several times, the input of MPRep, the stem of a head)."""
    if n <= 1:
        return [t]
    if not (t.is_cuda and t.requires_grad and not T.framework_ops):
        return [t] * n
    return list(T._Fanout.apply(t, n))


def fork(t, lo=0):
    """(t, t[:, lo:]) for a tensor that goes into a `join` AND (its channels lo..) into a later block: the block's gradient is added into the join's."""
    return T._Fork.apply(t, lo)


@T._laned
class _MaxPool(torch.autograd.Function):
    """MaxPool2d(k, stride, pad) on csrc/pool_train.hip: forward keeps a one-byte argmax, backward gathers (the framework's backward scatters
    with atomics over overlapping windows — 271 us per SPPF pool on 32 x 192 x 20 x 20 — and drags int64 indices along)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, out=None):
        x, xs = T.nhwc(x)
        B, c, H, W = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = T._empty((B, c, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if out is None else out[0]      # out: a concat buffer's slot (SPPF)
        idx = T._empty((B, Ho, Wo, c), dtype=torch.uint8, device=x.device)
        lib.check(lib.load().maf_maxpool_forward(x.data_ptr(), xs, B, H, W, c, k, stride, pad, T._DT[x.dtype], y.data_ptr(), y.stride()[3], idx.data_ptr(), T._stream(x.device)))
        ctx.save_for_backward(idx)
        ctx.geom = (H, W, k, stride, pad)
        T.stats["native_maxpool"] = T.stats.get("native_maxpool", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        H, W, k, stride, pad = ctx.geom
        dy, dys = T.nhwc(dy)
        B, c = dy.shape[:2]
        dx = T._empty((B, c, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        lib.check(lib.load().maf_maxpool_backward(dy.data_ptr(), dys, idx.data_ptr(), B, H, W, c, k, stride, pad, T._DT[dy.dtype], dx.data_ptr(), dx.stride()[3], T._stream(dy.device)))
        return dx, None, None, None, None


@T._laned
class _Up2(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode="nearest") on csrc/pool_train.hip: the source may be a channel slice (a concat buffer's slot), the result may go into one."""

    @staticmethod
    def forward(ctx, x, out):
        x, xs = T.nhwc(x)
        B, c, H, W = x.shape
        y = T._empty((B, c, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if out is None else out[0]
        lib.check(lib.load().maf_upsample2x_forward(x.data_ptr(), xs, B, H, W, c, T._DT[x.dtype], y.data_ptr(), y.stride()[3], T._stream(x.device)))
        T.stats["native_upsample"] = T.stats.get("native_upsample", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        dy, dys = T.nhwc(dy)
        B, c, H2, W2 = dy.shape
        dx = T._empty((B, c, H2 // 2, W2 // 2), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        lib.check(lib.load().maf_upsample2x_backward(dy.data_ptr(), dys, B, H2 // 2, W2 // 2, c, T._DT[dy.dtype], dx.data_ptr(), dx.stride()[3], T._stream(dy.device)))
        return dx, None


def upsample2x(x, out=None):
    """Nearest-neighbour x2 (the neck's nn.Upsample nodes); `out`: a concat buffer's slot or a callable that returns it for a [B, C, 2H, 2W] tensor like x.  CUDA
    fp16 / fp32 tensors with whole 16-byte channel groups run csrc/pool_train.hip, anything else the framework's kernel (out is then ignored: the concat copies)."""
    mult = 8 if x.dtype == torch.float16 else 4
    if not (x.is_cuda and not T.framework_ops and x.dtype in T._DT and x.dim() == 4 and x.shape[1] % mult == 0):
        return F.interpolate(x, scale_factor=2, mode="nearest")
    if out is not None:
        if callable(out):
            B, c, H, W = x.shape
            out = out(T.Like((B, c, 2 * H, 2 * W), x.dtype, x.device)) if T.cat_free else None
        if out is not None and not (out.dtype == x.dtype and out.device == x.device and tuple(out.shape) == (x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3]) and T.nhwc(out)[0] is out):
            raise lib.MafError("upsample2x: out= must be an NHWC (channel-slice) view of the result's shape and dtype")
    return T._Up2.apply(x, None if out is None else (out,))


def maxpool_native_ok(x, k, stride=1, pad=None):
    """maxpool(x, ...) would run csrc/pool_train.hip (and take `out=`)."""
    if pad is None:
        pad = k // 2 if stride == 1 else 0
    mult = 8 if x.dtype == torch.float16 else 4
    return not T.framework_ops and x.is_cuda and x.dtype in T._DT and x.dim() == 4 and x.shape[1] % mult == 0 and 2 <= k <= 15 and 1 <= stride <= k and 2 * pad <= k


def maxpool(x, k, stride=1, pad=None, out=None):
    """F.max_pool2d(x, k, stride, pad) (pad default k // 2 for stride 1, else 0) with autograd; CUDA fp16 / fp32 tensors with channels in whole
    16-byte groups run the HIP kernels.  `out`: a concat buffer's slot of the result's shape (HIP path only: ask `maxpool_native_ok` first)."""
    if pad is None:
        pad = k // 2 if stride == 1 else 0
    if not T.maxpool_native_ok(x, k, stride, pad):
        if out is not None:
            raise lib.MafError("maxpool: out= is a feature of the HIP path")
        if x.is_cuda:
            T.stats["torch_maxpool"] = T.stats.get("torch_maxpool", 0) + 1
        return F.max_pool2d(x, k, stride, pad)
    if out is not None:
        Ho, Wo = (x.shape[2] + 2 * pad - k) // stride + 1, (x.shape[3] + 2 * pad - k) // stride + 1
        if not (out.dtype == x.dtype and out.device == x.device and tuple(out.shape) == (x.shape[0], x.shape[1], Ho, Wo) and T.nhwc(out)[0] is out):
            raise lib.MafError("maxpool: out= must be an NHWC (channel-slice) view of the result's shape and dtype")
    return T._MaxPool.apply(x, k, stride, pad, None if out is None else (out,))


def maxpool_s1(x, k, out=None):
    return T.maxpool(x, k, 1, k // 2, out=out)
