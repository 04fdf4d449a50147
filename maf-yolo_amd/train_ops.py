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


@_laned
class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, bnslot=None):
        """bnslot: (bn, holder) — the training-mode BatchNorm2d behind the conv and an empty list: when the conv's tile has the statistics epilogue, the
        (scratch, phase) the BatchNorm call must be given as `pre_stats` is appended to the list."""
        x, xs = nhwc(x)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dt = _DT[x.dtype]
        co = -(-cout // 4) * 4                                                   # the kernel stores 4 channels at a time: any class count
        M = B * H * W
        want = bnslot is not None and conv_bn_stats and not _deterministic and bias is None and co == cout
        choice = _conv_tune.get((M, cin, co, xs, "st") if want else (M, cin, co, xs)) if conv_autotune and dt == lib.F16 else None
        w2d = None
        if choice is None:
            w2d = w.detach().reshape(cout, cin).float().contiguous()
            if co != cout:                                                       # (cls_pred with nc % 4 != 0) runs with zero filters appended
                w2d = F.pad(w2d, (0, 0, 0, co - cout))
            choice = _conv_choice(x, xs, B, H, W, cin, co, dt, w2d, co, cin, 0, want)
        pt, ct, tk = choice
        wp = _hit(w, ("d", co, cin, 1, 0, dt, ct)) if co == cout else None       # staged by this step's batch (PackPlan)
        if wp is None:
            if w2d is None:
                w2d = w.detach().reshape(cout, cin).float().contiguous()
                if co != cout:
                    w2d = F.pad(w2d, (0, 0, 0, co - cout))
            wp = _packed_1x1(w2d, co, cin, 0, dt, ct, x.device, w if co == cout else None)
        npad = -(-co // (16 * ct)) * 16 * ct
        if bias is None:
            bp = _zero_bias(x.device, npad)
        else:
            bp = _staged_bias(bias, cout, npad, x.device)                       # the bias on the conv's channel tile, zero behind it: staged by the step's pack batch
        out = _empty((B, co, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        bstat = None
        if bnslot is not None and _conv_stats_ok(choice, cin, co, cout, dt, bias):
            bstat = bn_own_scratch(bnslot[0], x.device, cout)
            bnslot[1].append(bstat)
            stats["conv_bn_stats"] = stats.get("conv_bn_stats", 0) + 1
        _launch_conv1x1(x, xs, wp, bp, B, H, W, cin, co, ct, out, dt, pt, tk, bstat)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None   # (an input of this node, not a saved tensor: backward adds its gradient straight into a gradient exchange)
        stats["native_conv1x1"] += 1
        return out if co == cout else out[:, :cout]

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = nhwc(dy)
        if dy.dtype != x.dtype:
            _glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = _DT[x.dtype]
        dx = dw = db = None
        mult = 8 if x.dtype == torch.float16 else 4
        dyk, dyks, kk = dy, dys, cout
        if cout % mult and (ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and x.dtype == torch.float16)):
            kk = -(-cout // mult) * mult                                         # e.g. reg_pred: 68 channels in fp16 — dY zero-padded to whole 16-byte chunks ONCE, for
            if dys >= kk and zero_padded.get(dy.data_ptr(), 0) >= kk:            # the weight gradient and the data gradient's reduction dim
                dyk = dy.as_strided((B, kk, H, W), dy.stride())                  # its producer keeps the padding channels zero: no copy
            else:
                _glue()
                dyk = F.pad(dy, (0, 0, 0, 0, 0, kk - cout)).contiguous(memory_format=torch.channels_last)
                dyks = kk
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ex, view = _grad_sink(ctx.bias_param) if ctx.bias_param is not None and x.dtype == torch.float16 else (None, None)
            if ex is not None and view.is_contiguous() and cout <= 256:
                # the bias gradient as a column sum on the weight-gradient stream, added into the bias' slice of the gradient exchange (csrc/train_ops.hip
                # maf_colsum) — a framework reduction + an accumulation add on the main stream otherwise
                h = _fork(x.device, dy)
                lib.check(lib.load().maf_colsum(dy.data_ptr(), dys, B * H * W, cout, dt, view.data_ptr(), h))
                ex.side_done(ctx.bias_param)
                stats["native_bias_grad"] = stats.get("native_bias_grad", 0) + 1
            else:
                db = dy.sum((0, 2, 3), dtype=torch.float32)
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:                                        # csrc/wgrad.hip: pixel chunks, LDS transpose, MFMA, fp32 atomics
                dw = _wgrad(x, dyk, dyks, w, 1, 1)                               # any Cin (channel chunks of 256), any Cout (dY padded to 8 channels); None: went into the exchange
            else:                                                                # fp32 parity mode: the framework's TN GEMM
                x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)                      # NHWC rows (a view when x is dense)
                d2 = dy.permute(0, 2, 3, 1).reshape(-1, cout)
                dw = torch.mm(d2.t(), x2).float().reshape(w.shape).to(w.dtype)
                stats["framework_wgrad_fp32"] = stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            # W^T: dX[m, ci] = sum_co dY[m, co] W[co, ci].  A dY padded to kk > cout channels needs no padded weight: the packer zero-fills K up to whole k-steps,
            # and rounding cout up to 8 never crosses one — the fragment record of [cout] rows IS the one of [kk] rows
            w2d = None
            M = B * H * W
            choice = _conv_tune.get((M, kk, cin, dyks)) if conv_autotune and dt == lib.F16 else None
            if choice is None:
                w2d = w.detach().reshape(cout, cin).float().contiguous()
                choice = _conv_choice(dyk, dyks, B, H, W, kk, cin, dt, w2d, cout, cin, 1)
            pt, ct, tk = choice
            wp = _hit(w, ("d", cout, cin, 1, 1, dt, ct))
            if wp is None:
                if w2d is None:
                    w2d = w.detach().reshape(cout, cin).float().contiguous()
                wp = _packed_1x1(w2d, cout, cin, 1, dt, ct, x.device, w)
            npad = -(-cin // (16 * ct)) * 16 * ct
            dx = _empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            _launch_conv1x1(dyk, dyks, wp, _zero_bias(x.device, npad), B, H, W, kk, cin, ct, dx, dt, pt, tk)
        _side_done(x.device, dw is not None)
        return dx, dw, db, None


def _wgrad(x, dy, dys, w, ksize, stride):
    """fp16 weight gradient on csrc/wgrad.hip (maf_conv_wgrad): x [B,Cin_x,Hs,Ws], dy [B,Cout,Ho,Wo] NHWC views -> dW like w, fp32 (Cin_x >= w's
    input channels: the stem's image padded to 8).  Launched on the side stream (`_fork`): call it BEFORE the data gradient of the layer is
    launched.  With a gradient exchange the result is accumulated into w's slice of its bucket on that stream and None is returned."""
    B, cin, Hs, Ws = x.shape
    cdy, Ho, Wo = dy.shape[1:]
    cout, cin_w = w.shape[0], w.shape[1]
    xx, xs = nhwc(x)
    co = -(-cdy // 8) * 8
    if co != cdy:                                                               # e.g. reg_pred: 68 channels, an odd class count (the 1x1 backward hands dY in padded already)
        _glue()                                                                 # (torch kernels: not recordable by a step tape)
        dy = F.pad(dy, (0, 0, 0, 0, 0, co - cdy)).contiguous(memory_format=torch.channels_last)
        dys = co
    ex, view = _grad_sink(w)
    direct = ex is not None and ksize == 1 and co == cout and cin == cin_w      # the kernel's [Cout][Cin] IS the parameter's layout: accumulate in place
    L = lib.load()
    if direct:
        dwf = view
        h = _fork(x.device, xx, dy)
    else:
        dwf = _empty((co, cin) if ksize == 1 else (3, 3, co, cin), dtype=torch.float32, device=x.device)       # 3x3: tap-major (csrc/wgrad.hip)
        h = _fork(x.device, xx, dy, dwf)
        lib.check(L.maf_zero(dwf.data_ptr(), dwf.numel() * 4, h))
    with _prof("conv_wgrad_k%d" % ksize, (B * Hs * Ws * cin + B * Ho * Wo * co) * 2 + dwf.numel() * 4, x.device, (B, Hs, Ws, cin, co, xs, dys, stride), h):
        lib.check(L.maf_conv_wgrad(xx.data_ptr(), xs, dy.data_ptr(), dys, B, Ho, Wo, Hs, Ws, cin, co, ksize, stride, lib.F16, dwf.data_ptr(), h))
    stats["native_wgrad"] = stats.get("native_wgrad", 0) + 1
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
    hit = _hit(w, ("d", cout, cin, 9, int(transpose), dt, ct))
    if hit is not None:
        return hit
    ks = 32 if dt == lib.F16 else 16
    n, k = (cin, cout) if transpose else (cout, cin)
    kp = -(-k // ks) * ks

    def now(dst):
        m = w.detach().float().permute(1, 2, 3, 0) if transpose else w.detach().float().permute(0, 2, 3, 1)      # [N, 3, 3, K]
        big = F.pad(m, (0, kp - k)).reshape(n, 9 * kp).contiguous()
        lib.check(lib.load().maf_pack_w1x1(big.data_ptr(), n, 9 * kp, 0, dt, ct, dst.data_ptr(), _stream(dev)))

    nbytes = lib.load().maf_pack_w1x1_bytes(n, 9 * kp, 0, dt, ct)
    if not (w.dtype == torch.float32 and w.is_contiguous() and w.is_leaf):        # a temporary (e.g. the stem's channel-padded filters): no plan entry
        buf = _empty(nbytes, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    fields = dict(kind=0, dtype=dt, Cout=cout, Cin=cin, taps=9, transpose=int(transpose), CT=ct, steps=9 * kp // ks, Kp=kp, flip=0,
                  total=nbytes // (2 if dt == lib.F16 else 4))
    return _staged(w, ("d", cout, cin, 9, int(transpose), dt, ct), nbytes, fields, now)


_conv3_tune = {}


def _conv3_choice(op, key, cands, w, transpose, dt, dev):
    """(tile_p, tile_c, tile_k) of a 3 x 3 stride-2 launch (forward: MAF_OP_CONV3X3S2, data gradient: MAF_OP_CONV3X3S2_DGRAD): like `_conv_choice`, every
    candidate is timed once per shape on the tensors at hand (the static rule — pack.tile_for / _tile_dgrad — left the neck's 128 -> 128 side convs at 1.2 TB/s)."""
    best = _conv3_tune.get(key)
    if best is not None:
        return best
    torch.cuda.synchronize(dev)
    timer, st, res = lib.Timer(), _stream(dev), []
    global profile, _plan
    saved, profile = profile, None
    saved_plan, _plan = _plan, None                                             # the candidates' weight forms are packed here and now: only the winner's joins the staging plan
    L = lib._lib if lib._lib is not None else lib.load()                        # (never through a recording tape's proxy)
    try:
        for pt, ct, tk in cands:
            n = op.Cout
            op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
            op.w, op.bias = _packed_3x3(w, transpose, dt, ct, dev).data_ptr(), _zero_bias(dev, -(-n // (16 * ct)) * 16 * ct).data_ptr()
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
        profile, _plan = saved, saved_plan
    res.sort()
    best = _conv3_tune[key] = res[0][1:] if res else cands[-1]
    stats["conv_tuned"] = stats.get("conv_tuned", 0) + 1
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


@_laned
class _Conv3x3s2(torch.autograd.Function):
    """nn.Conv2d(k=3, stride=2, padding=1, bias=False): forward csrc/conv_mfma.inc.h VAR_3X3S2, data gradient VAR_DGRAD3 (gather form),
    weight gradient csrc/wgrad.hip with the taps gathered in the kernel."""

    @staticmethod
    def forward(ctx, x, w):
        x, xs = nhwc(x)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dt = _DT[x.dtype]
        pt, ct = pack.tile_for(cout, B * Ho * Wo)
        tk = 1
        out = _empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        op = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV3X3S2, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout, op.nsrc = B, Ho, Wo, H, W, cin, cout, 1
        op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = x.data_ptr(), cin, xs, 0, lib.SRC_DIRECT
        op.out, op.out_stride, op.out_coff = out.data_ptr(), out.stride()[3], 0
        if conv3_autotune and dt == lib.F16 and w.dtype == torch.float32 and w.is_leaf:
            key = ("f", B * Ho * Wo, cin, cout, xs)
            ch = _conv3_tune.get(key)
            if ch is None and _rec is None:                                      # (never timed inside a recording step: the static tile then)
                ch = _conv3_choice(op, key, _conv3_fwd_cands(cin, cout, B * Ho * Wo, (pt, ct, 1)), w, False, dt, x.device)
            if ch is not None:
                pt, ct, tk = ch
        wp = _packed_3x3(w, False, dt, ct, x.device)
        op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
        op.w, op.bias = wp.data_ptr(), _zero_bias(x.device, -(-cout // (16 * ct)) * 16 * ct).data_ptr()
        es = x.element_size()
        with _prof("conv3x3s2", (B * H * W * cin + B * Ho * Wo * cout + 9 * cin * cout) * es, x.device, (B, H, W, cin, cout, pt, ct)):
            lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
        ctx.save_for_backward(x, w)
        stats["native_conv3x3s2"] = stats.get("native_conv3x3s2", 0) + 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = nhwc(dy)
        if dy.dtype != x.dtype:
            _glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        Ho, Wo = dy.shape[2:]
        dt = _DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:
                dw = _wgrad(x, dy, dys, w, 3, 2)
            else:                                                                # fp32 parity mode: the framework's kernel
                dw = torch.nn.grad.conv2d_weight(x[:, :w.shape[1]], w.shape, dy, stride=2, padding=1).to(w.dtype)
                stats["framework_wgrad_fp32"] = stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            # (an input whose channels were padded — the image — gets zeros in the padding: the packer pads W^T's rows to the channel tile)
            pt, ct = _tile_dgrad(cin, B * H * W)
            dx = _empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            op = lib.MafOp()
            op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV3X3S2_DGRAD, dt, dt, lib.ACT_NONE
            op.B, op.H, op.W, op.Hin, op.Win, op.Cin, op.Cout, op.nsrc = B, H, W, Ho, Wo, cout, cin, 1
            op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = dy.data_ptr(), cout, dys, 0, lib.SRC_DIRECT
            op.out, op.out_stride, op.out_coff = dx.data_ptr(), dx.stride()[3], 0
            if conv3_autotune and dt == lib.F16 and w.dtype == torch.float32 and w.is_leaf:
                key = ("d", B * H * W, cin, cout, dys)
                ch = _conv3_tune.get(key)
                if ch is None and _rec is None:
                    ch = _conv3_choice(op, key, _conv3_dgrad_cands(cin, B * H * W, (pt, ct, 0)), w, True, dt, x.device)
                if ch is not None:
                    pt, ct = ch[0], ch[1]
            wp = _packed_3x3(w, True, dt, ct, x.device)
            op.tile_p, op.tile_c, op.tile_k = pt, ct, 0
            op.w, op.bias = wp.data_ptr(), _zero_bias(x.device, -(-cin // (16 * ct)) * 16 * ct).data_ptr()
            es = x.element_size()
            with _prof("conv3x3s2_dgrad", (B * H * W * cin + B * Ho * Wo * cout + 9 * cin * cout) * es, x.device, (B, H, W, cin, cout, pt, ct)):
                lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
        _side_done(x.device, dw is not None)
        return dx, dw


@_laned
class _Conv1x1s2(torch.autograd.Function):
    """nn.Conv2d(k=1, stride=2, bias=False) (RepVGGBlock.rbr_1x1, common.py:203): the 1x1 kernel reading pixel (2y, 2x) of its source
    (MAF_SRC_SUB2); data gradient = the 1x1 data gradient scattered onto the even pixels; weight gradient csrc/wgrad.hip, one gathered tap."""

    @staticmethod
    def forward(ctx, x, w):
        x, xs = nhwc(x)
        B, cin, H, W = x.shape
        assert H % 2 == 0 and W % 2 == 0, "stride-2 1x1 conv: even input sides (images are multiples of 32)"
        cout = w.shape[0]
        Ho, Wo = H // 2, W // 2
        dt = _DT[x.dtype]
        pt, ct = pack.tile_for(cout, B * Ho * Wo)
        cin_w = w.shape[1]                                                       # < cin for the stem: the image's channels are padded to 8, the weight's K to whole k-steps by the packer
        wp = _packed_1x1(w.detach().reshape(cout, cin_w).float().contiguous(), cout, cin_w, 0, dt, ct, x.device, w)
        out = _empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        op = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_CONV1X1, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Cin, op.Cout, op.nsrc = B, Ho, Wo, cin, cout, 1
        op.src[0].ptr, op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = x.data_ptr(), cin, xs, 0, lib.SRC_SUB2
        op.out, op.out_stride, op.out_coff = out.data_ptr(), out.stride()[3], 0
        op.tile_p, op.tile_c = pt, ct
        op.w, op.bias = wp.data_ptr(), _zero_bias(x.device, -(-cout // (16 * ct)) * 16 * ct).data_ptr()
        with _prof("conv1x1", B * Ho * Wo * (cin + cout) * x.element_size(), x.device):
            lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
        ctx.save_for_backward(x, w)
        stats["native_conv1x1"] += 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dy, dys = nhwc(dy)
        if dy.dtype != x.dtype:
            _glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = _DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.float16:
                dw = _wgrad(x, dy, dys, w, 1, 2)
            else:
                xsub = x[:, :w.shape[1], ::2, ::2].permute(0, 2, 3, 1).reshape(-1, w.shape[1])
                dw = torch.mm(dy.permute(0, 2, 3, 1).reshape(-1, cout).t(), xsub).float().reshape(w.shape).to(w.dtype)
                stats["framework_wgrad_fp32"] = stats.get("framework_wgrad_fp32", 0) + 1
        if ctx.needs_input_grad[0]:
            cin_w = w.shape[1]                                                   # < cin: padded image channels get a zero gradient (W^T's rows are padded to the channel tile)
            ct = pack.tile_for(cin, B * (H // 2) * (W // 2))[1]
            wp = _packed_1x1(w.detach().reshape(cout, cin_w).float().contiguous(), cout, cin_w, 1, dt, ct, x.device, w)
            dxs = _empty((B, cin, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            _launch_conv1x1(dy, dys, wp, _zero_bias(x.device, -(-cin // (16 * ct)) * 16 * ct), B, H // 2, W // 2, cout, cin, ct, dxs, dt)
            if getattr(ctx, "compact", False):                                   # _RepVGGConvs adds it onto the 3x3 branch's data gradient itself
                dx = dxs
            else:
                _glue(2)
                dx = _empty((B, cin, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
                dx[:, :, ::2, ::2] = dxs
        _side_done(x.device, dw is not None)
        return dx, dw


class _Ctx:
    """What a Function's forward / backward use of their ctx, for calling them from another Function."""
    needs_input_grad = (True, True)

    def save_for_backward(self, *t):
        self.saved_tensors = t


@_laned
class _RepVGGConvs(torch.autograd.Function):
    """(conv3x3 s2 (x, w3), conv1x1 s2 (x, w1)) — the two branches of a RepVGGBlock (yolov6/layers/common.py:199-203) as ONE autograd node, so that their
    data gradients meet inside it: the 1x1 branch's gradient lives on the even pixels only and is added onto the 3x3 branch's in place (maf_add_sub2, a quarter
    of the pixels) instead of a zero-filled full-size tensor + a strided copy + autograd's full-size add."""

    @staticmethod
    def forward(ctx, x, w3, w1):
        if stem_train and x.dtype == torch.float16 and w3.shape[1] == 3 and x.shape[1] == 8 and w3.shape[0] % 8 == 0 and w3.shape[0] <= 96 \
                and w3.dtype == torch.float32 and w1.dtype == torch.float32 and w3.is_contiguous() and w1.is_contiguous() and x.shape[3] % 8 == 0:
            # the image (3 channels padded to 8): both branches in ONE launch of a direct conv (csrc/stem_train.hip) — the generic kernels read it twice with a K of 72 / 8
            x, xs = nhwc(x)
            B, _, H, W = x.shape
            cout = w3.shape[0]
            z3 = _empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            z1 = _empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            with _prof("stem_train", (B * H * W * 8 + 2 * B * (H // 2) * (W // 2) * cout) * 2, x.device, (B, H, W, cout)):
                lib.check(lib.load().maf_stem_train(x.data_ptr(), xs, B, H, W, w3.data_ptr(), w1.data_ptr(), cout, z3.data_ptr(), z1.data_ptr(), _stream(x.device)))
            ctx.save_for_backward(x, w3, w1)
            stats["native_conv3x3s2"] = stats.get("native_conv3x3s2", 0) + 1
            stats["native_conv1x1"] += 1
            stats["native_stem_train"] = stats.get("native_stem_train", 0) + 1
            return z3, z1
        c3, c1 = _Ctx(), _Ctx()
        z3 = _Conv3x3s2.forward(c3, x, w3)
        z1 = _Conv1x1s2.forward(c1, x, w1)
        ctx.save_for_backward(c3.saved_tensors[0], w3, w1)
        return z3, z1

    @staticmethod
    def backward(ctx, dz3, dz1):
        x, w3, w1 = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        c3, c1 = _Ctx(), _Ctx()
        c3.saved_tensors, c3.needs_input_grad = (x, w3), (need_x, ctx.needs_input_grad[1])
        c1.saved_tensors, c1.needs_input_grad, c1.compact = (x, w1), (need_x, ctx.needs_input_grad[2]), True
        dx, dw3 = _Conv3x3s2.backward(c3, dz3)
        dxs, dw1 = _Conv1x1s2.backward(c1, dz1)
        if need_x:
            B, c, Ho, Wo = dxs.shape
            lib.check(lib.load().maf_add_sub2(dxs.data_ptr(), dxs.stride()[3], dx.data_ptr(), dx.stride()[3], B, Ho, Wo, c, _DT[dx.dtype], _stream(dx.device)))
        return dx, dw3, dw1


def repvgg_convs(x, w3, w1):
    """(conv3x3s2(x, w3), conv1x1s2(x, w1)) of one input."""
    if not x.is_cuda or framework_ops or x.shape[2] % 2 or x.shape[3] % 2:
        return conv3x3s2(x, w3), conv1x1s2(x, w1)
    x = _autocast(x)
    if not (x.dtype in _DT and x.dim() == 4 and tuple(w3.shape[2:]) == (3, 3) and tuple(w1.shape[2:]) == (1, 1)):
        raise lib.MafError("repvgg_convs: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return _RepVGGConvs.apply(_pad8(x, w3), w3, w1)


def pad_channels8(x):
    """A 3-channel image for kernels that read 16-byte channel chunks: cast as a convolution would under autocast, zero channels appended.
    RepVGGBlock does it ONCE for its two branches (32 x 3 x 640 x 640: the cast and the padded copy cost 0.15 ms each)."""
    x = _autocast(x)
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
        x = pad_channels8(x)
    if x.shape[1] != -(-cin // 8) * 8:
        raise lib.MafError("conv: input has %d channels, the filters %d" % (x.shape[1], cin))
    return x


def conv3x3s2(x, w):
    """nn.Conv2d(k=3, stride=2, padding=1, bias=False) with autograd; x [B,Cin,H,W] (NHWC in memory preferred), w [Cout,Cin,3,3]."""
    if not x.is_cuda or framework_ops:      # CPU tensors: the train-form module tree in plain torch (CI / gloo tests only)
        stats["fallback"] += 1
        if framework_ops and x.shape[1] > w.shape[1]:      # RepVGGBlock hands the image zero-padded to 8 channels (pad_channels8)
            x = x[:, :w.shape[1]]
        return F.conv2d(x, w if framework_ops else w.to(x.dtype), None, 2, 1)
    x = _autocast(x)
    if not (x.dtype in _DT and x.dim() == 4 and tuple(w.shape[2:]) == (3, 3)):
        raise lib.MafError("conv3x3s2: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return _Conv3x3s2.apply(_pad8(x, w), w)


def conv1x1s2(x, w):
    """nn.Conv2d(k=1, stride=2, bias=False) with autograd."""
    if not x.is_cuda or framework_ops:
        stats["fallback"] += 1
        if framework_ops and x.shape[1] > w.shape[1]:
            x = x[:, :w.shape[1]]
        return F.conv2d(x, w if framework_ops else w.to(x.dtype), None, 2, 0)
    x = _autocast(x)
    if not (x.dtype in _DT and x.dim() == 4 and tuple(w.shape[2:]) == (1, 1)):
        raise lib.MafError("conv1x1s2: unsupported input for the HIP path: %s %s" % (tuple(x.shape), x.dtype))
    return _Conv1x1s2.apply(_pad8(x, w), w)


def conv1x1_bn(x, w, bn):
    """(conv1x1(x, w), pre_stats): the 1x1 conv in front of the BatchNorm2d `bn` (Conv.forward, yolov6/layers/common.py:44-47).  When `bn` normalises with batch
    statistics on the HIP path and the conv runs on the persistent LDS-weight kernel, the conv's epilogue accumulates them and `pre_stats` is what
    bn_act(..., pre_stats=) takes (apply pass only); otherwise None."""
    if not (x.is_cuda and bn.training and bn.affine) or framework_ops or not conv_bn_stats or _deterministic:
        return conv1x1(x, w), None
    x = _autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (_ok(x, mult) and w.shape[2] == 1):
        raise lib.MafError("conv1x1: unsupported input for the HIP path: %s %s -> %d channels" % (tuple(x.shape), x.dtype, w.shape[0]))
    holder = []
    z = _Conv1x1.apply(x, w, None, (bn, holder))
    return z, (holder[0] if holder else None)


def conv1x1(x, w, bias=None):
    """nn.Conv2d(k=1, stride=1) forward with autograd. x [B,Cin,H,W] (NHWC in memory preferred), w [Cout,Cin,1,1]."""
    if not x.is_cuda or framework_ops:      # CPU tensors: the train-form module tree in plain torch (CI / gloo tests only)
        stats["fallback"] += 1
        if framework_ops:                   # autocast (if any) casts the operands itself
            return F.conv2d(x, w, bias)
        return F.conv2d(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))
    x = _autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (_ok(x, mult) and w.shape[2] == 1):
        raise lib.MafError("conv1x1: unsupported input for the HIP path: %s %s -> %d channels" % (tuple(x.shape), x.dtype, w.shape[0]))
    return _Conv1x1.apply(x, w, bias)


def _launch_dw(x, xs, wp, bias, B, H, W, c, k, out, dt):
    ys = out.stride()[3]
    key = (2, dt, B, H, W, c, k, xs, ys)
    op = _op_cache.get(key)
    if op is None:
        op = _op_cache[key] = lib.MafOp()
        op.kind, op.dtype, op.in_dtype, op.act = lib.OP_DWCONV, dt, dt, lib.ACT_NONE
        op.B, op.H, op.W, op.Cin, op.Cout, op.ksize, op.nsrc = B, H, W, c, c, k, 1
        op.src[0].C, op.src[0].stride, op.src[0].coff, op.src[0].mode = c, xs, 0, lib.SRC_DIRECT
        op.out_stride, op.out_coff = ys, 0
    op.src[0].ptr, op.out, op.w, op.bias = x.data_ptr(), out.data_ptr(), wp.data_ptr(), bias.data_ptr()
    if profile is None:
        lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))
        return
    with _prof("dwconv_k%d" % k, 2 * B * H * W * c * x.element_size(), x.device):
        lib.check(lib.load().maf_op_launch(C.byref(op), _stream(x.device)))


def _packed_dw(w, c, k, flip, dt, dev):
    hit = _hit(w, ("w", c, k, flip, dt))
    if hit is not None:
        return hit
    nbytes = c * k * k * (2 if dt == lib.F16 else 4)

    def now(dst):
        wf = w.detach().reshape(c, k * k).float().contiguous()
        lib.check(lib.load().maf_pack_dw(wf.data_ptr(), c, k, flip, dt, dst.data_ptr(), _stream(dev)))

    if not (w.dtype == torch.float32 and w.is_contiguous() and w.is_leaf):
        buf = _empty(nbytes, dtype=torch.uint8, device=dev)
        now(buf)
        return buf
    fields = dict(kind=1, dtype=dt, Cout=c, Cin=1, taps=k * k, transpose=0, CT=0, steps=0, Kp=0, flip=flip, total=c * k * k)
    return _staged(w, ("w", c, k, flip, dt), nbytes, fields, now)


@_laned
class _DWConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, xs = nhwc(x)
        B, c, H, W = x.shape
        k = w.shape[-1]
        dt = _DT[x.dtype]
        out = _empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _launch_dw(x, xs, _packed_dw(w, c, k, 0, dt, x.device), _zero_bias(x.device, c), B, H, W, c, k, out, dt)
        ctx.save_for_backward(x, w)
        stats["native_dwconv"] += 1
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, c, H, W = x.shape
        k = w.shape[-1]
        dy, dys = nhwc(dy)
        if dy.dtype != x.dtype:
            _glue()
            dy = dy.to(x.dtype)
            dys = dy.stride()[3]
        dt = _DT[x.dtype]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            # one copy of dW: the kernel adds one value per (channel, tap) and workgroup after its own LDS reduction, so the replicas the
            # first version spread its atomics over (and the torch sum behind them) buy <= 7 % on the 160 x 160 layers and nothing elsewhere
            dw = _dw_wgrad(x, dy, dys, w)
        if ctx.needs_input_grad[0]:                                              # correlation with the flipped kernel
            dx = _empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            _launch_dw(dy, dys, _packed_dw(w, c, k, 1, dt, x.device), _zero_bias(x.device, c), B, H, W, c, k, dx, dt)
        _side_done(x.device, dw is not None)
        return dx, dw


def _dw_wgrad(x, dy, dys, w):
    """Weight gradient of a depth-wise conv on the side stream (csrc/train_ops.hip: dw_wgrad_kernel): None when it went into a gradient exchange's
    bucket slice, else dW like w."""
    B, c, H, W = x.shape
    k = w.shape[-1]
    dt = _DT[x.dtype]
    xx, xs = nhwc(x)
    ex, view = _grad_sink(w)
    L = lib.load()
    if ex is not None:                                                        # [C][k*k] is the parameter's layout: the atomics land in its bucket slice
        dwf = view
        h = _fork(x.device, xx, dy)
    else:
        dwf = _empty(c, k * k, dtype=torch.float32, device=x.device)
        h = _fork(x.device, xx, dy, dwf)
        lib.check(L.maf_zero(dwf.data_ptr(), dwf.numel() * 4, h))
    with _prof("dw_wgrad_k%d" % k, 2 * B * H * W * c * x.element_size(), x.device, (B, H, W, c, k, xs, dys), h):
        lib.check(L.maf_dw_wgrad(xx.data_ptr(), xs, dy.data_ptr(), dys, B, H, W, c, k, dt, dwf.data_ptr(), 1, h))
    if ex is not None:
        ex.side_done(w)
        return None
    return dwf.reshape(w.shape).to(w.dtype)


stem_train = cfg.stem_train                # A/B switch: the image's two RepVGG convs as one direct-conv launch (csrc/stem_train.hip)
dw_wgrad31 = cfg.dw_wgrad31                # A/B switch: the 3x3 (+ 3x3) + 1x1 branches' weight gradients as one launch (x staged once)


def _dw_wgrad31_ok(x, ws):
    """The branches the merged launch takes: kernel sizes (.., 3, 1) behind an optional larger first branch, on the maps where maf_dw_wgrad routes k = 3 to the
    vector kernel (csrc/train_ops.hip: maf_dw_wgrad; the small maps' k = 3 gradients run on the matrix cores)."""
    if not (dw_wgrad31 and x.is_cuda and x.dtype in _DT):
        return None
    ks = tuple(int(w.shape[-1]) for w in ws)
    B, c, H, W = x.shape
    if W <= 96 and (H * W <= 400 or (H * W <= 1600 and c <= 192)):
        return None
    if ks == (3, 3, 1):
        return (0, 1, 2)
    if len(ks) == 3 and ks[1:] == (3, 1):
        return (1, None, 2)
    return None


def _dw_wgrad31(x, dzs, ws, sel):
    """maf_dw_wgrad31 on the side stream for the branches `sel` = (3x3, second 3x3 or None, 1x1): [dW or None (went into an exchange bucket)] per selected branch."""
    B, c, H, W = x.shape
    dt = _DT[x.dtype]
    xx, xs = nhwc(x)
    L = lib.load()
    js = [j for j in sel if j is not None]
    sinks = {j: _grad_sink(ws[j]) for j in js}
    bufs = {}
    for j in js:
        if sinks[j][0] is not None:
            bufs[j] = sinks[j][1]
        else:
            k = ws[j].shape[-1]
            bufs[j] = _empty(c, k * k, dtype=torch.float32, device=x.device)
    own = [bufs[j] for j in js if sinks[j][0] is None]
    h = _fork(x.device, xx, *[dzs[j] for j in js], *own)
    for t in own:
        lib.check(L.maf_zero(t.data_ptr(), t.numel() * 4, h))
    a, b, one = sel
    with _prof("dw_wgrad_k31", (1 + len(js)) * B * H * W * c * x.element_size(), x.device, (B, H, W, c, len(js), xs), h):
        lib.check(L.maf_dw_wgrad31(xx.data_ptr(), xs, dzs[a].data_ptr(), dzs[a].stride()[3],
                                   None if b is None else dzs[b].data_ptr(), 0 if b is None else dzs[b].stride()[3],
                                   dzs[one].data_ptr(), dzs[one].stride()[3], B, H, W, c, dt,
                                   bufs[a].data_ptr(), None if b is None else bufs[b].data_ptr(), bufs[one].data_ptr(), 1, h))
    out = {}
    for j in js:
        if sinks[j][0] is not None:
            sinks[j][0].side_done(ws[j])
            out[j] = None
        else:
            out[j] = bufs[j].reshape(ws[j].shape).to(ws[j].dtype)
    stats["native_dw_wgrad31"] = stats.get("native_dw_wgrad31", 0) + 1
    return out


_PTR4 = C.c_void_p * 4
_INT4 = C.c_int32 * 4


def _launch_dwb(srcs, dsts, wps, k0, B, H, W, c, dt, dgrad, dev, bstats=None):
    nb = len(wps)
    sp, ss = _PTR4(*[t.data_ptr() for t in srcs]), _INT4(*[t.stride()[3] for t in srcs])
    dp, ds = _PTR4(*[t.data_ptr() for t in dsts]), _INT4(*[t.stride()[3] for t in dsts])
    wp = _PTR4(*[t.data_ptr() for t in wps])
    es = 2 if dt == lib.F16 else 4
    L = lib.load()
    with _prof("dw_branches_dgrad_k%d" % k0 if dgrad else "dw_branches_k%d" % k0, (nb + 1) * B * H * W * c * es, dev, (B, H, W, c, k0, nb)):
        if bstats is not None:                                                   # [(scratch, phase)] per branch: the half its BatchNorm call will read
            half = _BN_REPLICAS * 2 * (-(-c // 256) * 256)
            stp = _PTR4(*[0 if st is None else st[0].data_ptr() + 4 * st[1] * half for st in bstats])
            if _rec is not None:                                                 # the half alternates from replay to replay: the pointer words toggle between the two
                _rec.toggle_array(stp, [(j, st[0].data_ptr() ^ (st[0].data_ptr() + 4 * half), 8) for j, st in enumerate(bstats) if st is not None])
            lib.check(L.maf_dw_branches_stats(sp, ss, dp, ds, wp, nb, k0, B, H, W, c, dt, stp, L.maf_bn_replicas(c, _BN_REPLICAS), _stream(dev)))
        else:
            lib.check(L.maf_dw_branches(sp, ss, dp, ds, wp, nb, k0, B, H, W, c, dt, 1 if dgrad else 0, _stream(dev)))


@_laned
class _DWBranches(torch.autograd.Function):
    """The parallel depth-wise branches of a train-form DilatedReparamBlock (yolov6/layers/common.py:3024-3031) on csrc/dw_branches.hip: one launch
    computes every branch's convolution of the shared input, one launch their summed data gradient; the weight gradients stay per branch on the side stream."""

    @staticmethod
    def forward(ctx, x, bstats, *ws):
        x, xs = nhwc(x)
        B, c, H, W = x.shape
        dt = _DT[x.dtype]
        dev = x.device
        outs = [_empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last) for _ in ws]
        _launch_dwb([x], outs, [_packed_dw(w, c, w.shape[-1], 0, dt, dev) for w in ws], ws[0].shape[-1], B, H, W, c, dt, False, dev, bstats)
        ctx.save_for_backward(x, *ws)
        stats["native_dwconv"] += len(ws)
        stats["native_dw_branches"] = stats.get("native_dw_branches", 0) + 1
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, *ws = ctx.saved_tensors
        B, c, H, W = x.shape
        dt = _DT[x.dtype]
        dev = x.device
        dzs = []
        for dy in dys:
            if dy is None:                                                       # a branch nobody used (not in the reference's graph): zero gradient
                _glue()
                dy = _tzeros((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            dy, _ = nhwc(dy)
            dzs.append(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        dws = [None] * len(ws)
        returned = False
        merged = {}
        sel = _dw_wgrad31_ok(x, ws) if all(ctx.needs_input_grad[2:2 + len(ws)]) else None
        if sel is not None:
            merged = _dw_wgrad31(x, dzs, ws, sel)
        for j, w in enumerate(ws):
            if j in merged:
                dws[j] = merged[j]
            elif ctx.needs_input_grad[2 + j]:
                dws[j] = _dw_wgrad(x, dzs[j], dzs[j].stride()[3], w)
            returned = returned or dws[j] is not None
        dx = None
        if ctx.needs_input_grad[0]:                                              # sum over the branches of the correlation with the flipped kernel
            dx = _empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            _launch_dwb(dzs, [dx], [_packed_dw(w, c, w.shape[-1], 1, dt, dev) for w in ws], ws[0].shape[-1], B, H, W, c, dt, True, dev)
        _side_done(dev, returned)
        return (dx, None, *dws)


_DWB_SETS = {3: (3, 3, 1), 5: (5, 3, 1), 7: (7, 5, 3), 9: (9, 7, 5, 3)}     # lk_origin + dil_branch_kernels(k) (arch.py), common.py:2997-3008
dw_branches_merged = cfg.dw_branches       # A/B switch: one launch per direction for the branches of a DilatedReparamBlock


dw_branch_stats = cfg.dw_branch_stats      # A/B switch: the branches' BatchNorm statistics out of the depth-wise kernel's epilogue
_deterministic = False


_own_scratch = __import__("weakref").WeakKeyDictionary()                  # BatchNorm2d module -> [scratch, phase, channels]: outside the module (the reference pickles / deep-copies whole models)


def bn_own_scratch(bn, dev, c):
    """(scratch, phase) of a BatchNorm whose statistics are produced by ANOTHER kernel than its own call (the depth-wise kernel of csrc/dw_branches.hip):
    a buffer per module — the shared per-stream one alternates its halves call by call, and the apply pass of the call in front would clear the half this
    call's producer has just filled — whose halves alternate step by step (the apply pass clears the half of the step before, as always)."""
    if _rec is not None:                                                         # a recording step tape: a scratch of its own per call site, the phase a toggled word
        return _tzeros(2 * _BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    ent = _own_scratch.get(bn)
    if ent is None or ent[0].device != dev or ent[2] != c:
        ent = _own_scratch[bn] = [_tzeros(2 * _BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1, c]
    ent[1] ^= 1
    return ent[0], ent[1]


def dw_branches(x, ws, bns=None):
    """[depth-wise conv of x with w for w in ws] for the k > 1 branches of a DilatedReparamBlock (kernel sizes k0, k0 - 2, ... 3; k0 = 3: 3, 3): ONE
    launch forward and one for the summed data gradient on CUDA tensors (csrc/dw_branches.hip); any other combination runs branch by branch.
    `bns` (the BatchNorm2d behind every branch): in training mode the kernel also accumulates every branch's batch statistics; returns (outputs,
    [per-branch `stats` argument for bn_act, or None])."""
    ks = tuple(int(w.shape[-1]) for w in ws)
    none = [None] * len(ws)
    if x.is_cuda and not framework_ops and dw_branches_merged and len(ws) > 1 and _DWB_SETS.get(ks[0]) == ks:
        x = _autocast(x)
        mult = 8 if x.dtype == torch.float16 else 4
        if _ok(x, mult):
            bstats = None
            if bns is not None and dw_branch_stats and not _deterministic and all(bn.training and bn.affine for bn in bns):
                bstats = [bn_own_scratch(bn, x.device, x.shape[1]) for bn in bns]
            outs = list(_DWBranches.apply(x, bstats, *ws))
            return (outs, bstats or none) if bns is not None else outs
    outs = [dwconv(x, w) for w in ws]
    return (outs, none) if bns is not None else outs


_ACT = {None: lib.ACT_NONE, "none": lib.ACT_NONE, "relu": lib.ACT_RELU, "silu": lib.ACT_SILU}
_BN_REPLICAS = 16


_bn_scratch = {}


def _bn_part(dev, c):
    """(scratch, phase): [2][R][2][roundup(c,256)] fp32, zeroed when it is allocated; a BatchNorm call accumulates into half `phase` and clears
    the other one (csrc/bn_act.hip), so the phase alternates per call on a buffer — kernels on one stream are ordered, different streams get
    different buffers."""
    if _rec is not None:
        return _tzeros(2 * _BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    key = (dev.index, _stream(dev), -(-c // 256))
    ent = _bn_scratch.get(key)
    if ent is None:
        if len(_bn_scratch) > 64:
            _bn_scratch.clear()
        ent = _bn_scratch[key] = [_tzeros(2 * _BN_REPLICAS * 2 * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1]
    ent[1] ^= 1
    return ent[0], ent[1]


@_laned
class _BNAct(torch.autograd.Function):
    """act(BatchNorm2d(x) [+ residual]) in training mode on the HIP kernels of csrc/bn_act.hip (batch statistics, running-stat update)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act, counter=None, residual=None, pre_stats=None, out=None):
        x, xs = nhwc(x)
        B, c, H, W = x.shape
        dt = _DT[x.dtype]
        dev = x.device
        # out: the caller's slot of a concat buffer (an NHWC channel slice, cat_buffer below): the apply pass stores there and the cat never runs
        y = _empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last) if out is None else out[0]      # (a tuple: not an input of the autograd node)
        stat = _empty(2, c, dtype=torch.float32, device=dev)                 # save_mean, save_rstd
        sp = stat.data_ptr()                                                      # (pointer arithmetic: indexing a tensor costs ~2 us of host time, 4 per call)
        part, phase = _bn_part(dev, c) if pre_stats is None else pre_stats       # pre_stats: (scratch, phase) whose half the producer of x has filled
        g32 = gamma.detach() if gamma.dtype == torch.float32 and gamma.is_contiguous() else gamma.detach().float().contiguous()
        b32 = beta.detach() if beta.dtype == torch.float32 and beta.is_contiguous() else beta.detach().float().contiguous()
        rs = 0
        if residual is not None:
            residual, rs = nhwc(residual)
        npass = 3 if residual is None else 4
        with _prof("bn_act_forward", npass * B * H * W * c * x.element_size(), dev, (B, H, W, c, xs, act)):    # statistics pass (read) + apply pass (read [, read], write)
            lib.check(lib.load().maf_bn_forward_ex(x.data_ptr(), xs, B * H * W, c, dt, g32.data_ptr(), b32.data_ptr(), float(eps), float(momentum),
                                                None if running_mean is None else running_mean.data_ptr(),
                                                None if running_var is None else running_var.data_ptr(),
                                                None if counter is None else counter.data_ptr(), act,
                                                y.data_ptr(), y.stride()[3], sp, sp + 4 * c, part.data_ptr(), _BN_REPLICAS,
                                                   phase, None if residual is None else residual.data_ptr(), rs, 0 if pre_stats is None else 1, _stream(dev)))
        ctx.has_res = residual is not None
        ctx.res_in_bwd = residual is not None and act != lib.ACT_NONE          # the activation's derivative needs u = BN(x) + residual
        if ctx.res_in_bwd:
            ctx.save_for_backward(x, g32, b32, stat, residual)
        else:
            ctx.save_for_backward(x, g32, b32, stat)
        ctx.act = act
        # the Parameters themselves (not saved tensors: they are inputs of this node): backward adds dgamma / dbeta straight into their slices of a gradient exchange
        ctx.affine = (gamma, beta) if isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) else None
        stats["native_bn_act"] = stats.get("native_bn_act", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dz):
        if ctx.res_in_bwd:
            x, g32, b32, stat, residual = ctx.saved_tensors
        else:
            (x, g32, b32, stat), residual = ctx.saved_tensors, None
        B, c, H, W = x.shape
        dz, dzs = nhwc(dz)
        if dz.dtype != x.dtype:
            dz = dz.to(x.dtype)
            dzs = dz.stride()[3]
        x, xs = nhwc(x)
        dev = x.device
        dx = _empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        # with a gradient exchange the apply kernel ADDS dgamma / dbeta to the parameters' bucket slices (main stream) and autograd gets None:
        # no AccumulateGrad add kernel per affine parameter (280 launches per step of n)
        from . import exchange
        ex, tg, tb = exchange.current, None, None
        if ex is not None and ctx.affine is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and bn_affine_direct:
            tg, tb = ex.target(ctx.affine[0]), ex.target(ctx.affine[1])
        direct = tg is not None and tb is not None and tg[1].is_contiguous() and tb[1].is_contiguous()
        dgb = None if direct else _empty(2, c, dtype=torch.float32, device=dev)                  # dgamma, dbeta
        part, phase = _bn_part(dev, c)
        dres, rs = None, 0
        if residual is not None:
            residual, rs = nhwc(residual)
            dres = _empty((B, c, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        npass = 5 if residual is None else 8
        with _prof("bn_act_backward", npass * B * H * W * c * x.element_size(), dev, (B, H, W, c, xs, dzs, ctx.act)):      # reduction pass (x, dz read) + apply pass (x, dz read, dx written)
            lib.check(lib.load().maf_bn_backward_acc(x.data_ptr(), xs, dz.data_ptr(), dzs, B * H * W, c, _DT[x.dtype], g32.data_ptr(), b32.data_ptr(),
                                                     stat.data_ptr(), stat.data_ptr() + 4 * c, ctx.act, dx.data_ptr(), dx.stride()[3],
                                                     tg[1].data_ptr() if direct else dgb.data_ptr(), tb[1].data_ptr() if direct else dgb.data_ptr() + 4 * c,
                                                     part.data_ptr(), _BN_REPLICAS, phase,
                                                     None if residual is None else residual.data_ptr(), rs,
                                                     None if dres is None else dres.data_ptr(), 0 if dres is None else dres.stride()[3], 1 if direct else 0, _stream(dev)))
        if ctx.has_res and dres is None:
            dres = dz                                                            # no activation: the residual's gradient is dz itself
        if direct:
            ex.main_done(ctx.affine[0])
            ex.main_done(ctx.affine[1])
            return dx, None, None, None, None, None, None, None, None, dres, None, None
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, dres, None, None


def bn_act(x, bn, act=None, residual=None, pre_stats=None, out=None):
    """act(bn(x) [+ residual]) for an nn.BatchNorm2d `bn` and act in {None, 'relu', 'silu'}.  `out`: a slot of a concat buffer (`CatBuffer.slot`; HIP path only)
    the result is stored into — and returned as.  `pre_stats`: what dw_branches returned for this branch (its
    kernel has accumulated the batch statistics already: apply pass only), else None.  Training mode on CUDA tensors runs the fused HIP
    kernels (one statistics pass + one normalise/affine/[add]/activation pass; backward likewise); eval mode and CPU tensors run torch ops.
    `residual` (same shape as x; act None or 'relu'): the branch sums of RepVGGBlock / DilatedReparamBlock without a pass of their own."""
    mult = 8 if x.dtype == torch.float16 else 4
    if not (x.is_cuda and bn.training) or framework_ops:
        # CPU tensors (CI / gloo tests) and eval-mode BatchNorm inside a train-form forward (Model.forward(val_loss=True) never comes here:
        # it runs the deploy engine): torch ops, counted so that an A/B on `stats` cannot mistake them for the HIP path
        stats["torch_bn"] = stats.get("torch_bn", 0) + 1
        if out is not None:
            raise lib.MafError("bn_act: out= is a feature of the HIP path (the caller checks `cat_free_ok`)")
        y = bn(x)
        if residual is not None:
            y = y + residual
        return y if act in (None, "none") else (F.relu(y) if act == "relu" else F.silu(y))
    if not (x.dtype in _DT and x.dim() == 4 and x.shape[1] % mult == 0 and bn.affine):
        raise lib.MafError("bn_act: unsupported input for the HIP path: %s %s (channels must be a multiple of %d, affine BatchNorm)" % (tuple(x.shape), x.dtype, mult))
    if residual is not None:
        if act == "silu":
            raise lib.MafError("bn_act: a residual goes with act None or 'relu'")
        if residual.shape != x.shape:
            raise lib.MafError("bn_act: residual %s must have the shape of x %s" % (tuple(residual.shape), tuple(x.shape)))
        if residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
    counter = bn.num_batches_tracked if bn.track_running_stats else None        # += 1 inside the apply kernel (140 one-element launches per step otherwise)
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64):
        counter.add_(1)
        counter = None
    if bn.momentum is None and bn.track_running_stats:
        # nn.BatchNorm2d(momentum=None) = cumulative moving average (factor 1 / num_batches_tracked): not what the kernel implements, and
        # not what the reference builds (momentum 0.03, yolov6/utils/torch_utils.py:43-45) — refuse rather than freeze the statistics
        raise lib.MafError("bn_act: BatchNorm2d(momentum=None) (cumulative average) is not supported on the HIP path")
    momentum = 0.0 if bn.momentum is None else bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if out is not None and not (out.shape == x.shape and out.dtype == x.dtype and out.device == x.device and nhwc(out)[0] is out):
        raise lib.MafError("bn_act: out= must be an NHWC (channel-slice) view of x's shape and dtype")
    return _BNAct.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, momentum, _ACT[act], counter, residual, pre_stats, None if out is None else (out,))


# Concats without a copy.  torch.cat of the train-form graph's RepHDW is three strided copies forward (its inputs are slices: no batched kernel), and
# backward a zero-fill + copy per slice, an add where a tensor feeds the cat AND the next block, and the split's cat of gradients: ~0.9 ms of the n step.
# Instead the producers' apply passes store straight into their channel slots of ONE buffer (`bn_act(out=)`), `join` hands that buffer to the consumer
# as a tensor whose gradient comes back as slot views, and `fork` (a tensor that feeds the cat and a later block) adds the block's gradient INTO the
# slot of the cat's gradient.  The in-place add is safe for what these two are built for: the gradient buffer is the data gradient the consumer conv has
# just written, `join.backward` is its only reader, and the slot is not read again before the add (the block's backward, which produced the addend,
# ran on OTHER slots).
cat_free = cfg.cat_free


def cat_free_ok(x, bn):
    """The copy-free concat needs the HIP BatchNorm path for the producers."""
    return cat_free and x.is_cuda and bn.training and not framework_ops and x.dtype in _DT


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
        self.buf = _empty((B, self.offs[-1], H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)

    def slot(self, i, n=1):
        """Channels of slots i .. i + n - 1 as a tensor of its own on the buffer's storage — NOT a view of `buf` for autograd: a slot becomes the output of its
        producer's autograd node, and a view whose base is written through another view later (join's copies) is refused there."""
        b = self.buf
        t = _empty(0, dtype=b.dtype, device=b.device)
        t.set_(b.untyped_storage(), b.storage_offset() + self.offs[i], (b.shape[0], self.offs[i + n] - self.offs[i], b.shape[2], b.shape[3]), b.stride())
        return t


@_laned
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
                dst = _empty(0, dtype=b_.dtype, device=b_.device)
                dst.set_(b_.untyped_storage(), b_.storage_offset() + o, (b_.shape[0], p_.shape[1], b_.shape[2], b_.shape[3]), b_.stride())
                mult = 8 if b_.dtype == torch.float16 else 4
                if b_.is_cuda and not framework_ops and b_.dtype in _DT and p_.shape[1] % mult == 0 and o % mult == 0 and nhwc(p_)[0] is p_:
                    nhwc_sum([p_], dst)
                else:
                    _glue()
                    dst.copy_(p_)
                stats["cat_copied_parts"] = stats.get("cat_copied_parts", 0) + 1
        stats["cat_free"] = stats.get("cat_free", 0) + 1
        return cb.buf

    @staticmethod
    def backward(ctx, d):
        return (None,) + tuple(d[:, a:b] for a, b in zip(ctx.offs[:-1], ctx.offs[1:]))


def join(cb, parts):
    """The concat of `parts` along the channels in CatBuffer `cb`: parts their producer stored into their slot (`bn_act(out=cb.slot(i))`) cost nothing, any
    other part is copied into its slot."""
    if any(p_.dtype != cb.buf.dtype or p_.shape[0] != cb.buf.shape[0] or p_.shape[2:] != cb.buf.shape[2:] or p_.device != cb.buf.device for p_ in parts):
        if cb.buf.is_cuda:
            _glue()
        return torch.cat(parts, 1)                                               # (mixed dtypes promote: the framework's rule)
    return _Join.apply(cb, *parts)


@_laned
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
            _glue()                                                               # the zero fill is a torch kernel that would run at recording time only: a step tape refuses this step
            d_all = _tzeros((d_tail.shape[0], ctx.lo + d_tail.shape[1]) + tuple(d_tail.shape[2:]), dtype=d_tail.dtype, device=d_tail.device).contiguous(memory_format=torch.channels_last)
        if d_tail is not None:
            tgt = d_all[:, ctx.lo:]
            mult = 8 if d_all.dtype == torch.float16 else 4
            if (d_all.is_cuda and not framework_ops and d_all.dtype in _DT and d_tail.dtype == d_all.dtype and d_tail.shape[1] % mult == 0 and ctx.lo % mult == 0
                    and nhwc(tgt)[0] is tgt and nhwc(d_tail)[0] is d_tail):
                nhwc_sum([d_tail], tgt, accumulate=True)                          # one launch on NHWC views (csrc/train_ops.hip maf_nhwc_sum), recordable by a step tape
            else:
                _glue()
                tgt.add_(d_tail)
        return d_all, None


class _DetectJoin(torch.autograd.Function):
    """Detect_yaml's train branch (yolov6/models/yolo.py:333-354) + the head's class sigmoid (yolov6/layers/common.py:1332): per-level NHWC (logits, box distribution)
    maps -> (cls [B,A,nc] probabilities, reg [B,A,4*(reg_max+1)]) in ONE launch (csrc/detect_join.hip), and one launch back: d logits = d cls * y * (1 - y), d reg, into
    per-level gradient maps padded to the conv kernels' 16-byte channel group (pad channels written as zeros: no memset, no F.pad in front of the weight gradient).
    A step tape records both calls."""

    @staticmethod
    def forward(ctx, nl, *ts):
        cls_l, reg_l = [nhwc(t) for t in ts[0::2]], [nhwc(t) for t in ts[1::2]]
        t0 = cls_l[0][0]
        B, nc = t0.shape[:2]
        nr = reg_l[0][0].shape[1]
        hw = [t.shape[2] * t.shape[3] for t, _ in cls_l]
        A = sum(hw)
        cls = _empty((B, A, nc), dtype=t0.dtype, device=t0.device)
        reg = _empty((B, A, nr), dtype=t0.dtype, device=t0.device)
        P, I = C.c_void_p * nl, C.c_int32 * nl
        hwa = I(*hw)
        with _prof("detect_join", 2 * B * A * (nc + nr) * t0.element_size(), t0.device):
            lib.check(lib.load().maf_detect_join(P(*[t.data_ptr() for t, _ in cls_l]), I(*[s_ for _, s_ in cls_l]), P(*[t.data_ptr() for t, _ in reg_l]), I(*[s_ for _, s_ in reg_l]),
                                                 hwa, nl, B, nc, nr, _DT[t0.dtype], cls.data_ptr(), reg.data_ptr(), _stream(t0.device)))
        stats["native_detect_join"] = stats.get("native_detect_join", 0) + 1
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
                _glue()                                                           # (the loss kernels and a step tape's boundary hand over contiguous tensors of the head's dtype)
        d_cls = None if d_cls is None else d_cls.to(dt).contiguous()
        d_reg = None if d_reg is None else d_reg.to(dt).contiguous()
        mult = 8 if dt == torch.float16 else 4
        ncp, nrp = -(-nc // mult) * mult, -(-nr // mult) * mult
        outs, dc, dr = [], [], []
        for h, w in shapes:
            gc = _empty((B, ncp, h, w), dtype=dt, device=dev, memory_format=torch.channels_last)
            gr = _empty((B, nrp, h, w), dtype=dt, device=dev, memory_format=torch.channels_last)
            dc.append(gc); dr.append(gr)
            vc, vr = (gc[:, :nc] if ncp != nc else gc), (gr[:, :nr] if nrp != nr else gr)
            if ncp != nc:
                zero_padded[vc.data_ptr()] = ncp                                  # the channels behind the view are zeros (the kernel writes them): _wgrad / the data gradient read whole groups
            if nrp != nr:
                zero_padded[vr.data_ptr()] = nrp
            outs += [vc, vr]
        P, I = C.c_void_p * nl, C.c_int32 * nl
        with _prof("detect_join_backward", B * sum(hw) * (3 * nc + 2 * nr) * cls.element_size(), dev):
            lib.check(lib.load().maf_detect_join_backward(None if d_cls is None else d_cls.data_ptr(), None if d_reg is None else d_reg.data_ptr(), cls.data_ptr(), I(*hw), nl, B, nc, nr,
                                                          _DT[dt], P(*[t.data_ptr() for t in dc]), I(*[ncp] * nl), P(*[t.data_ptr() for t in dr]), I(*[nrp] * nl), ncp, nrp, _stream(dev)))
        stats["native_detect_join"] = stats.get("native_detect_join", 0) + 1
        if _keep is not None:
            _keep.extend([d_cls, d_reg])
        return (None,) + tuple(outs)


def detect_join(heads):
    """(cls [B,A,nc] class probabilities, reg [B,A,4*(reg_max+1)]) from the per-level (stem, class LOGITS, box distribution) of the heads: Detect_yaml's train branch
    (yolov6/models/yolo.py:333-354: flatten + permute + cat) with the class sigmoid of Head_DepthUni (yolov6/layers/common.py:1332) folded in.  HIP tensors: one launch
    (csrc/detect_join.hip); CPU tensors / `framework_ops`: the reference's torch ops."""
    cls_l, reg_l = [h[1] for h in heads], [h[2] for h in heads]
    t0 = cls_l[0]
    native = (t0.is_cuda and not framework_ops and len(heads) <= 4 and t0.shape[1] % 4 == 0 and reg_l[0].shape[1] % 4 == 0
              and all(t.dim() == 4 and t.dtype == t0.dtype and t.dtype in _DT for t in cls_l + reg_l))
    if not native:
        if t0.is_cuda and not framework_ops:
            _glue(); stats["fallback"] += 1
        cls = torch.cat([torch.sigmoid(c).flatten(2).permute(0, 2, 1) for c in cls_l], 1)
        reg = torch.cat([r.flatten(2).permute(0, 2, 1) for r in reg_l], 1)
        return cls, reg
    return _DetectJoin.apply(len(heads), *[t for pair in zip(cls_l, reg_l) for t in pair])


def nhwc_sum(srcs, dst, accumulate=False):
    """dst = [dst +] sum(srcs) on NHWC views of one shape and dtype (csrc/train_ops.hip maf_nhwc_sum: 1..4 sources, channel slices welcome)."""
    B, c, H, W = dst.shape
    n = len(srcs)
    lib.check(lib.load().maf_nhwc_sum(_PTR4(*[t.data_ptr() for t in srcs]), _INT4(*[t.stride()[3] for t in srcs]), n, dst.data_ptr(), dst.stride()[3],
                                      B * H * W, c, _DT[dst.dtype], 1 if accumulate else 0, _stream(dst.device)))
    stats["native_nhwc_sum"] = stats.get("native_nhwc_sum", 0) + 1


@_laned
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
        if (d0.is_cuda and not framework_ops and d0.dtype in _DT and d0.dim() == 4 and d0.shape[1] % mult == 0 and len(live) <= 4
                and all(d.dtype == d0.dtype and d.shape == d0.shape for d in live)):
            views = [nhwc(d)[0] for d in live]
            out = _empty(d0.shape, dtype=d0.dtype, device=d0.device, memory_format=torch.channels_last)
            nhwc_sum(views, out)
            return out, None
        _glue(len(live) - 1)
        out = live[0] + live[1]
        for d in live[2:]:
            out = out + d
        return out, None


def fanout(t, n):
    """[t] * n for a tensor with n consumers inside the train-form graph (a backbone map the neck reads several times, the input of MPRep, the stem of a head)."""
    if n <= 1:
        return [t]
    if not (t.is_cuda and t.requires_grad and not framework_ops):
        return [t] * n
    return list(_Fanout.apply(t, n))


def fork(t, lo=0):
    """(t, t[:, lo:]) for a tensor that goes into a `join` AND (its channels lo..) into a later block: the block's gradient is added into the join's."""
    return _Fork.apply(t, lo)


_bnsum_scratch = {}


def _phase_array(phases):
    arr = _INT4(*[int(p_) for p_ in phases])
    if _rec is not None:
        _rec.toggle_array(arr, [(j, 1, 4) for j, p_ in enumerate(phases) if isinstance(p_, lib.Phase)])
    return arr


def _bnsum_part(dev, c, nb):
    """(scratch, phase) of maf_bn_sum_backward: [2][R][1 + nb][roundup(c,256)] fp32 per (stream, width, branch count), halves alternating call by call."""
    if _rec is not None:
        return _tzeros(2 * _BN_REPLICAS * (1 + nb) * (-(-c // 256) * 256), dtype=torch.float32, device=dev), lib.Phase(0)
    key = (dev.index, _stream(dev), -(-c // 256), nb)
    ent = _bnsum_scratch.get(key)
    if ent is None:
        if len(_bnsum_scratch) > 64:
            _bnsum_scratch.clear()
        ent = _bnsum_scratch[key] = [_tzeros(2 * _BN_REPLICAS * (1 + nb) * (-(-c // 256) * 256), dtype=torch.float32, device=dev), 1]
    ent[1] ^= 1
    return ent[0], ent[1]


@_laned
class _BNSum(torch.autograd.Function):
    """sum_j BatchNorm2d_j(z_j) in training mode, no activation (the branch sum of a DilatedReparamBlock, yolov6/layers/common.py:3024-3031) on csrc/bn_sum.hip:
    ONE apply pass forward (the statistics come from the depth-wise kernel's epilogue or a statistics launch per branch that lacks them), one statistics + one
    apply launch backward for ALL branches (their upstream gradient is the same tensor)."""

    @staticmethod
    def forward(ctx, nb, cfg, *t):
        """t = z_0 .. z_{nb-1}, gamma_0 .. gamma_{nb-1}, beta_0 .. beta_{nb-1}; cfg = (eps, momentum, [(running_mean, running_var, counter, scratch, phase, need_stats)] per branch)"""
        zs = [nhwc(z) for z in t[:nb]]
        gammas, betas = t[nb:2 * nb], t[2 * nb:3 * nb]
        x0 = zs[0][0]
        B, c, H, W = x0.shape
        dt = _DT[x0.dtype]
        dev = x0.device
        eps, momentum, per, act, dst, nxt = cfg                                 # nxt: (scratch, phase) of the BatchNorm that normalises the sum next — this pass accumulates its statistics — or None
        L = lib.load()
        M_ = B * H * W
        for (z, zst), (rm, rv, cnt, part, phase, need) in zip(zs, per):
            if need:                                                             # this branch's producer has no statistics epilogue (the 1 x 1 scale branch)
                lib.check(L.maf_bn_stats(z.data_ptr(), zst, M_, c, dt, part.data_ptr(), _BN_REPLICAS, phase, _stream(dev)))
        out = _empty((B, c, H, W), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) if dst is None else dst      # dst: a concat buffer's slot (bn_act's out=)
        stat = _empty(nb, 2, c, dtype=torch.float32, device=dev)            # save_mean, save_rstd per branch
        sp = stat.data_ptr()
        g32 = [g.detach() if g.dtype == torch.float32 and g.is_contiguous() else g.detach().float().contiguous() for g in gammas]
        b32 = [b.detach() if b.dtype == torch.float32 and b.is_contiguous() else b.detach().float().contiguous() for b in betas]
        # (the argument arrays are built OUTSIDE the timed region: on a host-bound eager step the event pair would measure their construction)
        args = (_PTR4(*[z.data_ptr() for z, _ in zs]), _INT4(*[zst for _, zst in zs]), nb, M_, c, dt,
                _PTR4(*[g.data_ptr() for g in g32]), _PTR4(*[b.data_ptr() for b in b32]), float(eps), float(momentum),
                _PTR4(*[0 if p[0] is None else p[0].data_ptr() for p in per]), _PTR4(*[0 if p[1] is None else p[1].data_ptr() for p in per]),
                _PTR4(*[0 if p[2] is None else p[2].data_ptr() for p in per]),
                out.data_ptr(), out.stride()[3], _PTR4(*[sp + 8 * c * j for j in range(nb)]), _PTR4(*[sp + 8 * c * j + 4 * c for j in range(nb)]),
                _PTR4(*[p[3].data_ptr() for p in per]), _BN_REPLICAS, _phase_array([p[4] for p in per]), act, _stream(dev))
        with _prof("bn_sum_forward", (nb + 1) * M_ * c * x0.element_size(), dev, (B, H, W, c, nb)):
            if nxt is None:
                lib.check(L.maf_bn_sum_forward(*args))
            else:
                lib.check(L.maf_bn_sum_forward_stats(*args[:-1], nxt[0].data_ptr(), _BN_REPLICAS, nxt[1], args[-1]))
        ctx.save_for_backward(stat, *[z for z, _ in zs], *g32, *b32)
        ctx.nb, ctx.act = nb, act
        ctx.affine = list(zip(gammas, betas)) if all(isinstance(g, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter) for g, b in zip(gammas, betas)) else None
        stats["native_bn_act"] = stats.get("native_bn_act", 0) + nb
        stats["native_bn_sum"] = stats.get("native_bn_sum", 0) + 1
        return out

    @staticmethod
    def backward(ctx, dy):
        nb = ctx.nb
        sv = ctx.saved_tensors
        stat, zs, g32, b32 = sv[0], [nhwc(z) for z in sv[1:1 + nb]], sv[1 + nb:1 + 2 * nb], sv[1 + 2 * nb:1 + 3 * nb]
        x0 = zs[0][0]
        B, c, H, W = x0.shape
        dev = x0.device
        dy, dys = nhwc(dy)
        if dy.dtype != x0.dtype:
            dy = dy.to(x0.dtype)
            dys = dy.stride()[3]
        dzs = [_empty((B, c, H, W), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) for _ in range(nb)]
        from . import exchange
        ex, tg = exchange.current, None
        if ex is not None and ctx.affine is not None and bn_affine_direct and all(ctx.needs_input_grad[2 + nb + j] and ctx.needs_input_grad[2 + 2 * nb + j] for j in range(nb)):
            tg = [(ex.target(g), ex.target(b)) for g, b in ctx.affine]
            if not all(a_ is not None and b_ is not None and a_[1].is_contiguous() and b_[1].is_contiguous() for a_, b_ in tg):
                tg = None
        dgb = None if tg is not None else _empty(nb, 2, c, dtype=torch.float32, device=dev)
        part, phase = _bnsum_part(dev, c, nb)
        sp = stat.data_ptr()
        gp = None if dgb is None else dgb.data_ptr()
        args = (dy.data_ptr(), dys, _PTR4(*[z.data_ptr() for z, _ in zs]), _INT4(*[zst for _, zst in zs]), nb, B * H * W, c, _DT[x0.dtype],
                                                     _PTR4(*[g.data_ptr() for g in g32]), _PTR4(*[b.data_ptr() for b in b32]), _PTR4(*[sp + 8 * c * j for j in range(nb)]), _PTR4(*[sp + 8 * c * j + 4 * c for j in range(nb)]),
                                                     _PTR4(*[d.data_ptr() for d in dzs]), _INT4(*[d.stride()[3] for d in dzs]),
                                                     _PTR4(*[tg[j][0][1].data_ptr() if tg is not None else gp + 8 * c * j for j in range(nb)]),
                                                     _PTR4(*[tg[j][1][1].data_ptr() if tg is not None else gp + 8 * c * j + 4 * c for j in range(nb)]),
                                                     1 if tg is not None else 0, part.data_ptr(), _BN_REPLICAS, phase, ctx.act, _stream(dev))
        with _prof("bn_sum_backward", (2 + 3 * nb) * B * H * W * c * x0.element_size(), dev, (B, H, W, c, nb)):
            lib.check(lib.load().maf_bn_sum_backward(*args))
        if tg is not None:
            for g, b in ctx.affine:
                ex.main_done(g)
                ex.main_done(b)
            return (None, None, *dzs, *([None] * (2 * nb)))
        return (None, None, *dzs, *[dgb[j, 0] for j in range(nb)], *[dgb[j, 1] for j in range(nb)])


bn_sum_merged = cfg.bn_sum               # A/B switch: the branch BatchNorms of a DilatedReparamBlock as one apply pass per direction


bn_sum_next_stats = cfg.bn_sum_stats       # A/B switch: the sum's apply pass accumulates the statistics of the BatchNorm behind it


def bn_sum(zs, bns, pre_stats=None, act=None, out=None, next_bn=None):
    """act(sum_j bns[j](zs[j])): the branches of a train-form DilatedReparamBlock (act None) or of a RepVGGBlock (act "relu", common.py:224).  CUDA + training mode: csrc/bn_sum.hip (one apply pass forward,
    statistics + apply for all branches backward); otherwise — and for anything the kernel does not take — the chain of bn_act calls with `residual`.
    `pre_stats[j]`: what dw_branches returned for branch j (its statistics are already accumulated) or None.
    `next_bn`: the BatchNorm2d that normalises the result next (UniRepLKNetBlock.norm): returns (result, pre_stats for bn_act(result, next_bn, ...)) — the apply pass has
    accumulated that BatchNorm's batch statistics (csrc/bn_sum.hip, STATS form) — or (result, None) where it cannot."""
    nb = len(zs)
    pre = list(pre_stats) if pre_stats is not None else [None] * nb
    x = zs[0]
    mult = 8 if x.dtype == torch.float16 else 4
    ok = (bn_sum_merged and 2 <= nb <= 4 and x.is_cuda and not framework_ops and not _deterministic and x.dtype in _DT and x.dim() == 4 and x.shape[1] % mult == 0
          and all(z.shape == x.shape and z.dtype == x.dtype for z in zs)
          and all(bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.eps == bns[0].eps and bn.momentum == bns[0].momentum
                  and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64 for bn in bns))
    if act not in (None, "none", "relu"):
        raise lib.MafError("bn_sum: act must be None or 'relu'")
    if not ok:
        y = bn_act(zs[0], bns[0], pre_stats=pre[0])
        for j in range(1, nb):
            y = bn_act(zs[j], bns[j], act if j == nb - 1 else None, residual=y, pre_stats=pre[j], out=out if j == nb - 1 else None)
        return y if next_bn is None else (y, None)
    if out is not None and not (out.shape == x.shape and out.dtype == x.dtype and out.device == x.device and nhwc(out)[0] is out):
        raise lib.MafError("bn_sum: out= must be an NHWC (channel-slice) view of the branches' shape and dtype")
    per = []
    for bn, st in zip(bns, pre):
        part, phase = st if st is not None else bn_own_scratch(bn, x.device, x.shape[1])
        per.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, part, phase, st is None))
    nxt = None
    if (next_bn is not None and bn_sum_next_stats and next_bn.training and next_bn.affine and next_bn.track_running_stats and next_bn.momentum is not None
            and x.shape[1] // mult <= 256):
        nxt = bn_own_scratch(next_bn, x.device, x.shape[1])
        stats["bn_sum_next_stats"] = stats.get("bn_sum_next_stats", 0) + 1
    y = _BNSum.apply(nb, (bns[0].eps, bns[0].momentum, per, _ACT[act], out, nxt), *zs, *[bn.weight for bn in bns], *[bn.bias for bn in bns])
    return y if next_bn is None else (y, nxt)


@_laned
class _MaxPool(torch.autograd.Function):
    """MaxPool2d(k, stride, pad) on csrc/pool_train.hip: forward keeps a one-byte argmax, backward gathers (the framework's backward scatters
    with atomics over overlapping windows — 271 us per SPPF pool on 32 x 192 x 20 x 20 — and drags int64 indices along)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, out=None):
        x, xs = nhwc(x)
        B, c, H, W = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = _empty((B, c, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if out is None else out[0]      # out: a concat buffer's slot (SPPF)
        idx = _empty((B, Ho, Wo, c), dtype=torch.uint8, device=x.device)
        lib.check(lib.load().maf_maxpool_forward(x.data_ptr(), xs, B, H, W, c, k, stride, pad, _DT[x.dtype], y.data_ptr(), y.stride()[3], idx.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(idx)
        ctx.geom = (H, W, k, stride, pad)
        stats["native_maxpool"] = stats.get("native_maxpool", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        H, W, k, stride, pad = ctx.geom
        dy, dys = nhwc(dy)
        B, c = dy.shape[:2]
        dx = _empty((B, c, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        lib.check(lib.load().maf_maxpool_backward(dy.data_ptr(), dys, idx.data_ptr(), B, H, W, c, k, stride, pad, _DT[dy.dtype], dx.data_ptr(), dx.stride()[3], _stream(dy.device)))
        return dx, None, None, None, None


@_laned
class _Up2(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode="nearest") on csrc/pool_train.hip: the source may be a channel slice (a concat buffer's slot), the result may go into one."""

    @staticmethod
    def forward(ctx, x, out):
        x, xs = nhwc(x)
        B, c, H, W = x.shape
        y = _empty((B, c, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) if out is None else out[0]
        lib.check(lib.load().maf_upsample2x_forward(x.data_ptr(), xs, B, H, W, c, _DT[x.dtype], y.data_ptr(), y.stride()[3], _stream(x.device)))
        stats["native_upsample"] = stats.get("native_upsample", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        dy, dys = nhwc(dy)
        B, c, H2, W2 = dy.shape
        dx = _empty((B, c, H2 // 2, W2 // 2), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        lib.check(lib.load().maf_upsample2x_backward(dy.data_ptr(), dys, B, H2 // 2, W2 // 2, c, _DT[dy.dtype], dx.data_ptr(), dx.stride()[3], _stream(dy.device)))
        return dx, None


def upsample2x(x, out=None):
    """Nearest-neighbour x2 (the neck's nn.Upsample nodes); `out`: a concat buffer's slot or a callable that returns it for a [B, C, 2H, 2W] tensor like x.  CUDA
    fp16 / fp32 tensors with whole 16-byte channel groups run csrc/pool_train.hip, anything else the framework's kernel (out is then ignored: the concat copies)."""
    mult = 8 if x.dtype == torch.float16 else 4
    if not (x.is_cuda and not framework_ops and x.dtype in _DT and x.dim() == 4 and x.shape[1] % mult == 0):
        return F.interpolate(x, scale_factor=2, mode="nearest")
    if out is not None:
        if callable(out):
            B, c, H, W = x.shape
            out = out(Like((B, c, 2 * H, 2 * W), x.dtype, x.device)) if cat_free else None
        if out is not None and not (out.dtype == x.dtype and out.device == x.device and tuple(out.shape) == (x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3]) and nhwc(out)[0] is out):
            raise lib.MafError("upsample2x: out= must be an NHWC (channel-slice) view of the result's shape and dtype")
    return _Up2.apply(x, None if out is None else (out,))


def maxpool_native_ok(x, k, stride=1, pad=None):
    """maxpool(x, ...) would run csrc/pool_train.hip (and take `out=`)."""
    if pad is None:
        pad = k // 2 if stride == 1 else 0
    mult = 8 if x.dtype == torch.float16 else 4
    return not framework_ops and x.is_cuda and x.dtype in _DT and x.dim() == 4 and x.shape[1] % mult == 0 and 2 <= k <= 15 and 1 <= stride <= k and 2 * pad <= k


def maxpool(x, k, stride=1, pad=None, out=None):
    """F.max_pool2d(x, k, stride, pad) (pad default k // 2 for stride 1, else 0) with autograd; CUDA fp16 / fp32 tensors with channels in whole
    16-byte groups run the HIP kernels.  `out`: a concat buffer's slot of the result's shape (HIP path only: ask `maxpool_native_ok` first)."""
    if pad is None:
        pad = k // 2 if stride == 1 else 0
    if not maxpool_native_ok(x, k, stride, pad):
        if out is not None:
            raise lib.MafError("maxpool: out= is a feature of the HIP path")
        if x.is_cuda:
            stats["torch_maxpool"] = stats.get("torch_maxpool", 0) + 1
        return F.max_pool2d(x, k, stride, pad)
    if out is not None:
        Ho, Wo = (x.shape[2] + 2 * pad - k) // stride + 1, (x.shape[3] + 2 * pad - k) // stride + 1
        if not (out.dtype == x.dtype and out.device == x.device and tuple(out.shape) == (x.shape[0], x.shape[1], Ho, Wo) and nhwc(out)[0] is out):
            raise lib.MafError("maxpool: out= must be an NHWC (channel-slice) view of the result's shape and dtype")
    return _MaxPool.apply(x, k, stride, pad, None if out is None else (out,))


def maxpool_s1(x, k, out=None):
    return maxpool(x, k, 1, k // 2, out=out)


def dwconv(x, w):
    """Depth-wise k x k stride-1 'same' conv (groups == channels) with autograd. w [C,1,k,k], k in {3,5,7,9}."""
    k = w.shape[-1]
    if k == 1:                               # a 1x1 depth-wise conv is a per-channel scale
        return x * w.reshape(1, -1, 1, 1).to(x.dtype)
    if not x.is_cuda or framework_ops:       # CPU tensors: plain torch (CI / gloo tests only)
        stats["fallback"] += 1
        return F.conv2d(x, w if framework_ops else w.to(x.dtype), None, 1, k // 2, 1, x.shape[1])
    x = _autocast(x)
    mult = 8 if x.dtype == torch.float16 else 4
    if not (_ok(x, mult) and k in (3, 5, 7, 9)):
        raise lib.MafError("dwconv: unsupported input for the HIP path: %s %s k=%d" % (tuple(x.shape), x.dtype, k))
    return _DWConv.apply(x, w)
