"""Autograd functions that run the training-form 1x1 and depth-wise convolutions on the HIP kernels.

Train step of the reference: `Trainer.train_in_steps` (yolov6/core/engine.py:141-167) — forward under
autocast, scaled backward, DDP all-reduce of the fp32 grads.  The train-form graph keeps conv and BN
apart, so the convs here have no epilogue; BN / activations / pooling / cat stay torch ops on
channels_last tensors (NHWC in memory — exactly the layout the kernels take, so nothing is copied).

    conv1x1(x, w, bias=None)   forward + data gradient: csrc/conv_mfma.inc.h (dgrad = same kernel on W^T, packed on
                               the device by maf_pack_w1x1); weight gradient dW = dY^T X: csrc/wgrad.hip (fp16; fp32: torch.mm)
    dwconv(x, w)               forward + data gradient: csrc/dwconv.hip (dgrad = flipped kernel, maf_pack_dw);
                               weight gradient: csrc/train_ops.hip:dw_wgrad_kernel

Both accept fp16 or fp32 NHWC *views* (channel slices of wider buffers are fine) and fp32 master
weights; outputs are channels_last tensors of the input dtype, weight grads are fp32.
CUDA tensors always take the HIP kernels (unsupported shapes raise); CPU tensors run plain torch ops so the
train-form module tree can be exercised by the CPU/gloo tests (`stats` counts which path ran)."""
import ctypes as C

import os

import torch
import torch.nn.functional as F

from .config import cfg
from . import lib, pack

_DT = {torch.float16: lib.F16, torch.float32: lib.F32}
_zeros = {}
stats = {"native_conv1x1": 0, "native_dwconv": 0, "fallback": 0}

# Per-kernel timing of one training step (bench.py --train: the `roofline` object): while `profile` is a dict every native launch is
# bracketed by HIP events on its stream; profile_collect() turns them into {kind: [milliseconds, algorithmic bytes, launches]}.
profile = None
# A/B and parity tests only (never set by the product path): every entry point below takes its plain-torch branch on CUDA tensors too — the same
# module tree on the framework's convolutions / BatchNorm / pooling (MIOpen, under whatever autocast the caller set).  tests/test_gpu_train.py
# measures the autocast RECIPE with it, so that the HIP kernels are bounded against the recipe's own deviation from fp32.
framework_ops = False


# A step tape that is recording (tape.py) keeps every buffer a recorded launch may touch alive: the replays use the recorded addresses.
_keep = None
_rec = None                  # the recording tape itself (BatchNorm scratches get a buffer per call site and a toggled phase word)


def _empty(*a, **k):
    t = torch.empty(*a, **k)
    if _keep is not None:
        _keep.append(t)
    return t


_in_alloc = False             # (the fill kernel of a zeroed allocation is not glue: tape.py's debug check skips it)


def _tzeros(*a, **k):
    global _in_alloc
    _in_alloc = True
    try:
        t = torch.zeros(*a, **k)
    finally:
        _in_alloc = False
    if _keep is not None:
        _keep.append(t)
    if _cur_lane and t.is_cuda:
        # torch's fill kernel went to the MAIN stream, the kernels that are about to use the buffer go to lane _cur_lane, which forked from the main stream
        # BEFORE this fill was queued: without a wait the lane may accumulate into the buffer before (or while) it is zeroed.  Seen as a non-finite loss in about
        # every second run of two processes sharing one GPU (tests/test_gpu_train.py::test_bench_train_two_ranks_on_one_device — a BatchNorm scratch of a
        # recording step tape, allocated per call site inside a lane); a process that has the GPU to itself wins the race.  Host-side wait, not a tape record:
        # replays never allocate.
        lib.check(lib._lib.maf_stream_fork(_lane_handle(t.device, 0), _lane_handle(t.device, _cur_lane)))
    return t


def _glue(n=1):
    """A torch kernel ran where the HIP path has none (an unusual layout, a fall-back branch): counted — a step tape refuses a step that contains one."""
    stats["glue"] = stats.get("glue", 0) + n


def set_deterministic(on=True):
    """Bit-reproducible BatchNorm statistics (csrc/bn_act.hip, maf_set_deterministic): a test / debugging mode — three launches per BatchNorm pass and
    per-workgroup slots instead of atomics.  The forward pass of the train-form graph is then bit-identical from run to run (the weight-gradient
    kernels keep their fp32 atomics: continuous round-off only)."""
    global _deterministic
    lib.check(lib.load().maf_set_deterministic(1 if on else 0))
    _deterministic = bool(on)
profile_detail = None                # a list: profile_collect() also appends (kind, note, ms, bytes) per launch
_pending = []


class _NoProf:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOPROF = _NoProf()


def _prof(kind, nbytes, dev, note=None, stream=None):
    """Context around one native launch: HIP events while `profile` is a dict, nothing at all otherwise (~600 launches per step go through here)."""
    return _NOPROF if profile is None else _Prof(kind, nbytes, dev, note, stream)


class _Prof:
    def __init__(self, kind, nbytes, dev, note=None, stream=None):
        self.kind, self.nbytes, self.dev, self.note, self.stream = kind, int(nbytes), dev, note, stream

    def __enter__(self):
        if profile is not None:
            self.t = lib.Timer()
            self.t.start(self.stream if self.stream is not None else _stream(self.dev))

    def __exit__(self, *exc):
        if profile is not None:
            self.t.stop(self.stream if self.stream is not None else _stream(self.dev))
            _pending.append((self.kind, self.nbytes, self.t, self.note))
        return False


def profile_collect():
    """Wait for the recorded launches and fold them into `profile` (kind -> [ms, algorithmic bytes, launches])."""
    for kind, nbytes, t, note in _pending:
        rec = profile.setdefault(kind, [0.0, 0, 0])
        ms = t.elapsed_ms()
        rec[0] += ms; rec[1] += nbytes; rec[2] += 1
        if profile_detail is not None:
            profile_detail.append((kind, note, ms, nbytes))
    del _pending[:]
    return profile


_raw_stream = torch._C._cuda_getCurrentRawStream if hasattr(torch._C, "_cuda_getCurrentRawStream") else None


# Lanes: independent branches of the train-form graph (the class / box branch of every detection head, common.py:1288-1336) on streams of their own while a
# step tape records — and therefore in every replay.  Their kernels are small (20 x 20 ... 80 x 80 maps, 5-50 us each, bound by launch latency and fixed
# costs, not by the chip), so two or three such chains side by side cost little more than one.  Only a recording step uses them: the tape keeps every buffer
# alive, so memory handed from one stream to another needs no allocator bookkeeping; eager steps run everything on the current stream.
n_lanes = cfg.train_lanes      # at most (lane_handles: four hardware queues for main, weight gradients, lanes and RCCL)
_cur_lane = 0                                    # 0: the current (main) stream; k: lane k
_lane_streams = {}
_lane_rejects = []


def lane_handles(dev):
    """Raw handles of the lane streams of `dev`, created on first use.  The runtime multiplexes a process's streams onto FOUR hardware queues and streams that
    share one serialise — with cross-stream waits between them a step then takes 35 ms instead of 20 (measured twice: three lanes; two lanes beside RCCL's own
    stream) — so the lanes are few (main + weight-gradient stream + lanes [+ RCCL's stream] <= 4) and PROBED: a set is taken only if spin kernels on the main
    stream, the weight-gradient stream and the lanes at once take about as long as one alone (streams.overlap_ratio); else one lane fewer, down to none."""
    if n_lanes <= 0:
        return []
    ls = _lane_streams.get(dev.index)
    if ls is None:
        from .streams import concurrent_streams, overlap_ratio
        want = min(n_lanes, 2)
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                want = min(want, 1)
        except Exception:
            pass
        main, side = torch.cuda.current_stream(dev), side_stream(dev)
        picked = []
        while want > 0 and not picked:
            for _ in range(3):
                cand = concurrent_streams(dev, want)
                if overlap_ratio([main, side] + cand) < 1.5:
                    picked = cand
                    break
                _lane_rejects.append(cand)                                       # kept alive: a destroyed stream's queue slot would be handed out again
            want -= 1
        ls = _lane_streams[dev.index] = picked
    return [s_.cuda_stream for s_ in ls]


def join_lanes(dev):
    """The current stream waits for every lane of `dev` (host-side: the gradient exchange calls it before a bucket's all-reduce is ordered behind the main stream)."""
    ls = _lane_streams.get(dev.index)
    if ls:
        main = _lane_handle(dev, 0)
        for s_ in ls:
            lib.check(lib._lib.maf_stream_fork(s_.cuda_stream, main))


def lanes_on(dev):
    return _rec is not None and n_lanes > 0 and dev.type == "cuda" and bool(_lane_streams.get(dev.index))


def _stream(dev):
    """Raw handle of the device's current stream (torch.cuda.current_stream builds a Stream object per call: 4.5 us, ~600 calls per step)."""
    if _cur_lane:
        return _lane_streams[dev.index][_cur_lane - 1].cuda_stream
    if _raw_stream is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


# Weight gradients run on a SIDE stream: dW of a layer depends on nothing that follows it in the backward chain (dY -> dX -> BatchNorm
# backward -> ...), whose kernels are short and leave the chip idle while they ramp up and drain.  The side stream waits for the main
# stream at the launch (dY is ready).  Who waits for the side stream depends on where the gradient goes:
#   * with a gradient exchange (exchange.GradExchange, `exchange.current`): the kernel accumulates straight into the parameter's slice of a
#     flat bucket, the autograd Function returns None for the weight, nothing on the main stream reads dW, and the main stream waits ONCE at
#     the end of backward (GradExchange.finish) — the bucket all-reduces are launched from the side stream too;
#   * without one (plain autograd, DistributedDataParallel): the Function returns dW as a tensor that AccumulateGrad / the pad backward of the
#     stem / the DDP reducer read on the MAIN stream straight away, so the main stream waits for the side stream before the layer's backward
#     returns (`_side_done`) — the overlap is then with the data gradient of the same layer only.
wgrad_stream = cfg.wgrad_stream
bn_affine_direct = cfg.bn_affine_direct     # A/B switch: BatchNorm dgamma / dbeta into the exchange's bucket slices by the apply kernel
_side_streams = {}
_side_events = {}
_side_used = {}                                  # device index -> the side stream holds work the main stream has not waited for


def side_stream(dev):
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(dev)
        _side_events[dev.index] = torch.cuda.Event()
    return side


def join_side(dev):
    """The main (current) stream of `dev` waits for everything its side stream holds."""
    side = _side_streams.get(dev.index)
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
        _side_used[dev.index] = False


def _fork(dev, *tensors):
    """Raw handle of the stream a weight gradient is launched on: the device's side stream, made to wait for what the main stream has issued
    so far (dY is ready) — or the main stream itself when wgrad_stream is off.  `tensors` (allocated on the main stream) are kept from being
    reused before the side stream is done.  torch's current stream does not change (a torch.cuda.stream() context costs ~15 us per layer):
    the caller passes the handle to the C-ABI."""
    if not wgrad_stream:
        return _stream(dev)
    side = side_stream(dev)
    h = side.cuda_stream
    lib.check(lib.load().maf_stream_fork(_stream(dev), h))                       # event record on the main stream + wait on the side stream, by raw handles (a step tape records it)
    if _keep is None:                                                            # (a recording tape keeps every buffer alive itself)
        for t in tensors:
            t.record_stream(side)
    _side_used[dev.index] = True
    return h


def _side_done(dev, returned_dw):
    """End of a layer's backward.  `returned_dw`: the weight gradient goes back to autograd as a tensor — its consumers run on the main
    stream, which therefore waits for the side stream here."""
    if returned_dw and wgrad_stream and _side_used.get(dev.index):
        join_side(dev)


def _grad_sink(w):
    """(exchange, bucket view) when the weight gradient of parameter `w` goes straight into a gradient exchange, else (None, None)."""
    from . import exchange
    ex = exchange.current
    if ex is None or not isinstance(w, torch.nn.Parameter):
        return None, None
    ent = ex.target(w)
    if ent is None:
        return None, None
    return ex, ent[1]


def _zero_bias(dev, n):
    z = _zeros.get(dev.index)
    if z is None or z.numel() < n:
        z = _zeros[dev.index] = _tzeros(max(n, 4096), dtype=torch.float32, device=dev)
    return z


def nhwc(t):
    """(tensor, pixel stride in elements) with t's memory being an NHWC view; copies only if it is not."""
    B, Cc, H, W = t.shape
    s = t.stride()
    if s[1] == 1 and s[3] >= Cc and s[2] == W * s[3] and s[0] == H * W * s[3]:
        return t, s[3]
    if t.is_cuda:
        _glue()
    t = t.contiguous(memory_format=torch.channels_last)
    return t, t.stride()[3]


def _autocast(x):
    """Convolutions are on autocast's lower-precision list: under torch.autocast an fp32 input (e.g. what nn.Upsample
    returns, which then promotes the following cat) is cast down exactly as F.conv2d would do."""
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if x.dtype != dt and x.is_floating_point():
            return x.to(dt)
    return x


def _ok(x, cin_mult):
    return x.is_cuda and x.dtype in _DT and x.dim() == 4 and x.shape[1] % cin_mult == 0


_op_cache = {}                         # launch descriptors by geometry: only the pointers change from call to call (a ctypes field store is ~0.2 us)


def _launch_conv1x1(x, xs, wp, bias, B, H, W, cin, cout, ct, out, dt, pt=None, tk=1, bstat=None):
    """bstat: (scratch, phase) of the training-mode BatchNorm behind the conv (bn_own_scratch) — the conv's epilogue accumulates its batch statistics
    (csrc/conv_stream_lds_st.hip; the caller has checked `_conv_stats_ok` for the tile)."""
    ys = out.stride()[3]
    if pt is None:
        pt = pack.tile_for(cout, B * H * W)[0]
    key = (1, dt, B, H, W, cin, cout, pt, ct, tk, xs, ys)
    op = _op_cache.get(key)
    if op is None:
        op = _op_cache[key] = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV1X1, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Cin, op.Cout, op.nsrc = B, H, W, cin, cout, 1
        op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = cin, xs, 0, lib.SRC_DIRECT
        op.out_stride, op.out_coff = ys, 0
        op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
    op.src[0].ptr, op.out, op.w, op.bias = x.data_ptr(), out.data_ptr(), wp.data_ptr(), bias.data_ptr()
    op.aux[2], op.reserved0, op._tape_toggles = None, 0, None
    if bstat is not None:
        half = _BN_REPLICAS * 2 * (-(-cout // 256) * 256)
        p0 = bstat[0].data_ptr()
        op.aux[2], op.reserved0 = p0 + 4 * bstat[1] * half, lib.load().maf_bn_replicas(cout, _BN_REPLICAS)
        if _rec is not None:                                                     # the half alternates from replay to replay: the tape toggles the word in ITS copy of the descriptor
            op._tape_toggles = [(lib.MafOp.aux.offset + 2 * C.sizeof(C.c_void_p), p0 ^ (p0 + 4 * half), 8)]
    try:
        if profile is None:
            lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
            return
        es = x.element_size()
        with _prof("conv1x1", B * H * W * (cin + cout) * es + cin * cout * es, x.device):
            lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
    finally:
        if bstat is not None:
            op.aux[2], op.reserved0, op._tape_toggles = None, 0, None


conv_bn_stats = cfg.conv_bn_stats           # A/B switch: BatchNorm statistics out of the 1x1 conv's epilogue (csrc/conv_stream_lds_st.hip)


def _conv_stats_ok(choice, cin, co, cout, dt, bias):
    pt, ct, tk = choice
    return (conv_bn_stats and not _deterministic and tk == 5 and pt == 1 and dt == lib.F16 and bias is None and co == cout
            and bool(lib.load().maf_conv1x1_stats_supported(-(-cin // 32), ct)))


# Tile / variant choice of the training 1x1 convs (forward and data gradient): the first time a shape (pixels, K, N) is seen every candidate
# the inference tuner would try for it (engine.Plan.autotune: tile_p x tile_c of the generic kernel, split-K, LDS-shared weight fragments,
# the persistent "stream" forms) is timed on the tensors at hand and the best one kept — the static rule (pack.tile_for) loses 10-40 % on
# individual layers.  MAF_TRAIN_TUNE=0 turns it off.
conv_autotune = cfg.train_tune
conv3_autotune = conv_autotune and cfg.train_tune3        # the 3 x 3 stride-2 launches (forward, data gradient) alone
_conv_tune = {}


def _stream_lds_ok(ksteps, ct):
    from .engine import stream_lds_ok
    return stream_lds_ok(ksteps, ct)


def _conv_choice(x, xs, B, H, W, K, Nc, dt, w2d, rows, cols, transpose, want_stats=False):
    """(tile_p, tile_c, tile_k) for the single-source conv K -> Nc over x (w2d [rows][cols] as maf_pack_w1x1 takes it).  want_stats: the conv feeds a training-mode
    BatchNorm — a tile with the statistics epilogue (csrc/conv_stream_lds_st.hip) may cost what the statistics pass it removes would (launch + one read of the
    output) more than the fastest tile and still be chosen; kept under a key of its own (a data-gradient conv of the same shape has no use for it)."""
    M = B * H * W
    pt0, ct0 = pack.tile_for(Nc, M)
    if not conv_autotune or dt != lib.F16 or not x.is_cuda:
        return pt0, ct0, 1
    key = (M, K, Nc, xs, "st") if want_stats else (M, K, Nc, xs)
    best = _conv_tune.get(key)
    if best is not None:
        return best
    ksteps = -(-K // 32)
    cands = []
    for ct in (2, 4, 6, 8):
        nt = -(-Nc // (16 * ct))
        if nt * 16 * ct > 2 * max(Nc, 32) or (ct == 8 and Nc % 8):
            continue
        for pt in (1, 2, 4):
            if (pt == 4 and ct > 4) or (pt > 1 and -(-M // (64 * pt)) * nt < 256):
                continue
            cands.append((pt, ct, 1))
        if ksteps >= 8 and M <= 65536:
            cands.append((1, ct, 4))
        if ksteps <= 4 and ksteps * ct <= 16:
            cands += [(1, ct, 3), (2, ct, 3)]
        if _stream_lds_ok(ksteps, ct):
            cands.append((1, ct, 5))
            if ct >= 4 and 64 <= ksteps * ct <= 160 and (8 <= ksteps <= 20 or ksteps == 24):
                cands.append((2, ct, 5))                                         # eight waves behind one LDS copy of the weights (csrc/conv_stream_lds_w8.hip)
        if ksteps >= 4 and ct >= 4:
            for pt in ((1, 2, 4) if ct == 4 else (1, 2)):
                if pt == 1 or -(-M // (64 * pt)) * nt >= 256:
                    cands.append((pt, ct, 2))
                    if pt <= 2 and ksteps >= 8:
                        cands.append((pt, ct, 8))                                # ... by DMA, two k-steps per barrier (csrc/conv_mfma_dma.hip)
    if (pt0, ct0, 1) not in cands:
        cands.append((pt0, ct0, 1))
    out = _empty((B, Nc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    torch.cuda.synchronize(x.device)                                            # a quiet chip: the side stream's weight gradients would be in the timings
    timer, st, res, packs = lib.Timer(), _stream(x.device), [], {}
    global profile
    saved, profile = profile, None
    try:
        for pt, ct, tk in cands:
            if ct not in packs:
                packs[ct] = _packed_1x1(w2d, rows, cols, transpose, dt, ct, x.device)
            bp = _zero_bias(x.device, -(-Nc // (16 * ct)) * 16 * ct)
            try:
                _launch_conv1x1(x, xs, packs[ct], bp, B, H, W, K, Nc, ct, out, dt, pt, tk)          # warm-up (and validity)
            except lib.MafError:
                continue
            ts = []
            for _ in range(3):
                timer.start(st)
                _launch_conv1x1(x, xs, packs[ct], bp, B, H, W, K, Nc, ct, out, dt, pt, tk)
                timer.stop(st)
                ts.append(timer.elapsed_ms())
            res.append((min(ts), pt, ct, tk))
    finally:
        profile = saved
    res.sort()
    best = res[0][1:] if res else (pt0, ct0, 1)
    if want_stats and res:
        allow = 0.006 + M * Nc * 2 / 5.0e9                                       # ms: a statistics launch alone, tools/bn_bench.py (launch + bytes / 5 TB/s)
        elig = [r for r in res if _conv_stats_ok(r[1:], K, Nc, Nc, dt, None)]
        if elig and elig[0][0] <= res[0][0] + allow:
            best = elig[0][1:]
    _conv_tune[key] = best
    stats["conv_tuned"] = stats.get("conv_tuned", 0) + 1
    return best


class PackPlan:
    """Weight staging plan of one training model (model.py creates one per Model and calls `begin_step` at the start of every train-form
    forward).  A step needs every dense weight in MFMA fragment order twice (forward; transposed for the data gradient) and every depth-wise
    kernel twice (as is; flipped): ~250 pack launches of a few microseconds when issued per layer.  The plan remembers each transform the
    layers asked for (source parameter, geometry, a persistent destination) in a descriptor table on the device, and `begin_step` runs ALL
    of them in one launch (csrc/train_ops.hip maf_pack_batch).  A layer gets the staged buffer when the parameter's version counter still
    is the one the batch saw; otherwise — first step, weights edited since, layers called without a Model — it packs by itself, as before."""

    def __init__(self):
        self.entries = {}                    # key -> [param, dst, desc fields, version at the last pack]
        self.table = None
        self.nblocks = 0
        self.dirty = False

    def clear(self):
        self.__init__()


_plan = None


def begin_step(plan, dev):
    """Make `plan` the current one and stage every weight it knows in one launch on the current stream of `dev`."""
    global _plan
    _plan = plan
    from . import exchange
    if exchange.current is not None:                                            # a new forward/backward pass of the gradient exchange: per-pass state reset
        exchange.current.begin()
    if plan is None or not plan.entries or dev.type != "cuda":
        return
    ents = list(plan.entries.values())
    if plan.dirty or plan.table is None:
        arr = (lib.MafPackDesc * len(ents))()
        blk = 0
        for d, e in zip(arr, ents):
            f = e[2]
            d.src, d.dst, d.total = e[0].data_ptr(), e[1].data_ptr(), f["total"]
            d.kind, d.dtype, d.Cout, d.Cin, d.taps, d.transpose = f["kind"], f["dtype"], f["Cout"], f["Cin"], f["taps"], f["transpose"]
            d.CT, d.steps, d.Kp, d.flip, d.block0 = f["CT"], f["steps"], f["Kp"], f["flip"], blk
            blk += -(-f["total"] // 1024)
        assert C.sizeof(lib.MafPackDesc) == lib.load().maf_pack_desc_size()
        plan.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        plan.nblocks = blk
        plan.dirty = False
    lib.check(lib.load().maf_pack_batch(plan.table.data_ptr(), len(ents), plan.nblocks, _stream(dev)))
    for e in ents:
        e[3] = e[0]._version
    stats["pack_batches"] = stats.get("pack_batches", 0) + 1


def _hit(param, key):
    """The staged buffer of transform `key` of `param` if this step's batch filled it (the fast path of every layer call), else None."""
    plan = _plan
    if plan is None:
        return None
    e = plan.entries.get((param.data_ptr(),) + key)
    if e is not None and e[3] == param._version:                              # (the entry keeps its source alive: same address == same storage)
        return e[1]
    return None


def _staged(param, key, nbytes, fields, pack_now):
    """The packed form `key` of `param`: the plan's buffer when the batch of this step filled it, else pack_now(dst) into it (and remember
    the transform for the next step); without a plan a fresh buffer."""
    plan = _plan
    if plan is None:
        dst = _empty(nbytes, dtype=torch.uint8, device=param.device)
        pack_now(dst)
        return dst
    k = (param.data_ptr(),) + key
    e = plan.entries.get(k)
    if e is not None and e[3] == param._version:
        return e[1]
    if e is None:
        if len(plan.entries) >= 4096:                                            # not a model's fixed set of parameters: start over
            plan.clear()
        e = plan.entries[k] = [param, _empty(nbytes, dtype=torch.uint8, device=param.device), fields, -1]
        plan.dirty = True
    pack_now(e[1])
    e[3] = -1                                                                    # valid for this call only: the batch sets the version
    return e[1]


def _packed_1x1(w2d, cout, cin, transpose, dt, ct, dev, param=None):
    """Fragment-packed [cout][cin] weight (transpose: its transpose, the data gradient's operand).  `param`: the parameter w2d is a view of
    (same storage, fp32) — then the transform goes through the staging plan."""
    if param is not None:
        hit = _hit(param, ("d", cout, cin, 1, transpose, dt, ct))
        if hit is not None:
            return hit
    L = lib.load()
    n = L.maf_pack_w1x1_bytes(cout, cin, transpose, dt, ct)

    def now(dst):
        lib.check(L.maf_pack_w1x1(w2d.data_ptr(), cout, cin, transpose, dt, ct, dst.data_ptr(), _stream(dev)))

    if param is None or w2d.data_ptr() != param.data_ptr() or param.dtype != torch.float32 or not param.is_leaf:
        buf = _empty(n, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    ks = 32 if dt == lib.F16 else 16
    steps = -(-(cout if transpose else cin) // ks)
    fields = dict(kind=0, dtype=dt, Cout=cout, Cin=cin, taps=1, transpose=transpose, CT=ct, steps=steps, Kp=steps * ks, flip=0,
                  total=n // (2 if dt == lib.F16 else 4))
    return _staged(param, ("d", cout, cin, 1, transpose, dt, ct), n, fields, now)


def _staged_bias(bias, cout, npad, dev):
    """fp32 [npad]: the bias of a prediction conv followed by zeros (the conv reads its whole channel tile).  A parameter's copy is one more transform of the step's
    pack batch (kind 2); anything else is padded here."""
    def now(dst):
        d = dst.view(torch.float32)
        d.zero_()
        d[:cout].copy_(bias.detach())

    if not (isinstance(bias, torch.nn.Parameter) and bias.dtype == torch.float32 and bias.is_contiguous() and bias.is_leaf):
        buf = _empty(npad * 4, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    hit = _hit(bias, ("b", cout, npad))
    if hit is not None:
        return hit
    fields = dict(kind=2, dtype=lib.F32, Cout=cout, Cin=1, taps=1, transpose=0, CT=0, steps=0, Kp=0, flip=0, total=npad)
    return _staged(bias, ("b", cout, npad), npad * 4, fields, now)


# data_ptr -> channels: gradient buffers whose channels [C, that many) are kept zero by their producer (a step tape's boundary pads the 68 channels of a
# reg_pred gradient to 72 once): _Conv1x1.backward reads them as they are instead of padding a copy
zero_padded = {}


def _laned(cls):
    """Class decorator of the autograd Functions below: forward remembers the lane it ran on, backward issues its kernels on the same one."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a, **k):
        ctx._lane = _cur_lane
        return fwd(ctx, *a, **k)

    def backward(ctx, *g):
        global _cur_lane
        old, _cur_lane = _cur_lane, getattr(ctx, "_lane", _cur_lane)
        try:
            return bwd(ctx, *g)
        finally:
            _cur_lane = old

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _lane_handle(dev, k):
    if k == 0:
        global _cur_lane
        old, _cur_lane = _cur_lane, 0
        try:
            return _stream(dev)
        finally:
            _cur_lane = old
    return _lane_streams[dev.index][k - 1].cuda_stream


class _LaneSwitch(torch.autograd.Function):
    """Identity that hands a tensor from stream `src` to stream `dst` (0: the main stream, k: lane k): forward, `dst` waits for what `src` holds; backward, `src`
    waits for what `dst` holds (the gradient comes the other way)."""

    @staticmethod
    def forward(ctx, x, src, dst):
        ctx.src, ctx.dst, ctx.dev = src, dst, x.device
        lib.check(lib.load().maf_stream_fork(_lane_handle(x.device, src), _lane_handle(x.device, dst)))
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        lib.check(lib.load().maf_stream_fork(_lane_handle(ctx.dev, ctx.dst), _lane_handle(ctx.dev, ctx.src)))
        return g, None, None


def lane_run(k, fn, x):
    """fn(x) with its kernels on lane k (k = 0 or no recording tape: plain fn(x)).  The result lives on that lane: `lane_join` before anything on the main stream reads it."""
    global _cur_lane
    if k <= 0 or _cur_lane or not isinstance(x, torch.Tensor) or not lanes_on(x.device):
        return fn(x), 0
    k = k % (len(_lane_streams[x.device.index]) + 1)                             # chains 1, 2, 3 ... go round lane 1 .. lane n and the main stream (0)
    if k == 0:
        return fn(x), 0
    x = _LaneSwitch.apply(x, 0, k)
    _cur_lane = k
    try:
        y = fn(x)
    finally:
        _cur_lane = 0
    return y, k


def lane_join(y, k):
    """The main stream waits for lane k (what `lane_run` returned beside y); backward: the lane waits for the main stream's gradient."""
    return y if k <= 0 else _LaneSwitch.apply(y, k, 0)


_conv3_tune = {}


stem_train = cfg.stem_train                # A/B switch: the image's two RepVGG convs as one direct-conv launch (csrc/stem_train.hip)
dw_wgrad31 = cfg.dw_wgrad31                # A/B switch: the 3x3 (+ 3x3) + 1x1 branches' weight gradients as one launch (x staged once)


_PTR4 = C.c_void_p * 4
_INT4 = C.c_int32 * 4


_DWB_SETS = {3: (3, 3, 1), 5: (5, 3, 1), 7: (7, 5, 3), 9: (9, 7, 5, 3)}     # lk_origin + dil_branch_kernels(k) (arch.py), common.py:2997-3008
dw_branches_merged = cfg.dw_branches       # A/B switch: one launch per direction for the branches of a DilatedReparamBlock


dw_branch_stats = cfg.dw_branch_stats      # A/B switch: the branches' BatchNorm statistics out of the depth-wise kernel's epilogue
_deterministic = False


_own_scratch = __import__("weakref").WeakKeyDictionary()                  # BatchNorm2d module -> [scratch, phase, channels]: outside the module (the reference pickles / deep-copies whole models)


_ACT = {None: lib.ACT_NONE, "none": lib.ACT_NONE, "relu": lib.ACT_RELU, "silu": lib.ACT_SILU}
_BN_REPLICAS = 16


_bn_scratch = {}


# Concats without a copy.  torch.cat of the train-form graph's RepHDW is three strided copies forward (its inputs are slices: no batched kernel), and
# backward a zero-fill + copy per slice, an add where a tensor feeds the cat AND the next block, and the split's cat of gradients: ~0.9 ms of the n step.
# Instead the producers' apply passes store straight into their channel slots of ONE buffer (`bn_act(out=)`), `join` hands that buffer to the consumer
# as a tensor whose gradient comes back as slot views, and `fork` (a tensor that feeds the cat and a later block) adds the block's gradient INTO the
# slot of the cat's gradient.  The in-place add is safe for what these two are built for: the gradient buffer is the data gradient the consumer conv has
# just written, `join.backward` is its only reader, and the slot is not read again before the add (the block's backward, which produced the addend,
# ran on OTHER slots).
cat_free = cfg.cat_free


_bnsum_scratch = {}


bn_sum_merged = cfg.bn_sum               # A/B switch: the branch BatchNorms of a DilatedReparamBlock as one apply pass per direction


bn_sum_next_stats = cfg.bn_sum_stats       # A/B switch: the sum's apply pass accumulates the statistics of the BatchNorm behind it


# ---- the families (round 6: this file was 2 100 lines).  They reference everything above as `T.<name>` at call time and are re-exported here whole — private helpers too:
# tests, tools and tape.py reach them as train_ops.<name>, and replacing an entry point on this module (bench.py --torch-convs) still reaches every internal caller.
from . import train_conv, train_dw, train_bn, train_cat      # noqa: E402

for _m in (train_conv, train_dw, train_bn, train_cat):
    for _k, _v in vars(_m).items():
        if getattr(_v, "__module__", None) == _m.__name__:
            globals()[_k] = _v
del _m, _k, _v
