"""Step tape: the train-form forward and backward of one model as two recorded launch lists, replayed by one C call each.

Reference step: `Trainer.train_in_steps` (yolov6/core/engine.py:141-167) — `preds = self.model(images)` under autocast, the loss,
`self.scaler.scale(total_loss).backward()`.  The train-form graph here is ~420 kernel launches forward and ~470 backward (every conv,
BatchNorm, pool and concat slot of layers.py on its HIP kernel), 5-60 us each; issued one by one from Python autograd Functions the host
needs 19-27 ms per step — as long as the step takes on the device, so the GPU waits for Python on any box with a slower host core.

A `StepTape` removes the Python from the steady state without a second implementation of the graph:

* **record** — the third train-mode forward of a (batch shape, autocast dtype, gradient exchange) runs the normal way (layers.py -> train_ops.py ->
  libmafyolo_hip through ctypes) while `lib.load()` hands out a proxy that notes every tape-able C-ABI call beside making it: entry point, argument
  words, stream (main / weight-gradient side stream).  Every buffer such a call touches is kept alive by the tape (`train_ops._keep`), BatchNorm
  scratches become one buffer per call site whose phase word alternates between replays (`lib.Phase`, maf_tape_toggle), the fork of the weight-gradient
  stream is a C-ABI call too (maf_stream_fork).  The backward pass of that step is recorded the same way, between a boundary autograd node on the head
  outputs (which copies the six incoming gradients into static buffers: the recorded kernels read those) and the end of the engine's pass.
* **replay** — later steps copy the image into the static input, call `maf_tape_run` for the forward list, hand out aliases of the static head
  outputs from ONE autograd node; its backward copies the six gradients in and calls `maf_tape_run` for the backward list, cut at the points where the
  gradient exchange launched a bucket's all-reduce during the recording (exchange.py: the collectives go out from the same places).

What a tape cannot hold are torch kernels between the recorded calls (they would run once, at recording time): the train-form graph has none left on
the HIP path — concat copies, gradient sums of multi-consumer tensors and in-place adds are `maf_nhwc_sum` launches, the bias of a prediction conv is
staged by the step's pack batch — `train_ops.stats["glue"]` counts every fall-back branch that would run one (a tape refuses a step that did), and
`with tape.check():` proves it for a recording with a TorchDispatchMode that lists every device op seen inside the recorded regions
(tests/test_gpu_tape.py asserts the list is empty and that replayed steps follow the eager trajectory).

Contract: a recording is PROCESS-wide — `lib._recorder`, `train_ops._rec` and the current lane are module globals, so while the one recorded step
of a batch shape runs (its forward on the calling thread, its backward on the autograd engine's device thread) no other thread may issue library calls (an async
evaluation, a background NMS): they would be written into the tape.  Replays have no such restriction.  The outputs of a replayed forward are views of the tape's
static buffers, valid until the next forward of that shape: clone what must outlive the step.  `Model.eval()` / `invalidate()` release the tapes and what they pin.

Not for: fp32 parity runs, deterministic mode, profiling (`train_ops.profile`), plain autograd without a GradExchange (the weight gradients need their
static bucket slices), two forwards before a backward (while the first one's graph is alive; a forward whose graph was dropped
without a backward is released: StepTape.drop_pending) — all of these run the eager path, which stays the reference implementation of the step."""
import ctypes as C
import struct
import weakref

import torch

from .config import cfg
from . import lib, train_ops

_M64 = (1 << 64) - 1
_A_OFF = lib.MafTapeRec.a.offset
_REC = C.sizeof(lib.MafTapeRec)
RECORD_AT = 3                 # the n-th forward of a key is the recorded one: the first ones time the conv variants and fill the weight staging plan

_VIEW_OPS = ("aten.view.", "aten._unsafe_view.", "aten.as_strided.", "aten.slice.", "aten.select.", "aten.detach.", "aten.alias.", "aten.t.", "aten.transpose.",
             "aten.permute.", "aten.expand.", "aten.unsqueeze.", "aten.squeeze.", "aten.empty.", "aten.empty_strided.", "aten.empty_like.", "aten.set_.",
             "aten.reshape.", "aten._reshape_alias.", "aten.view_as.", "aten.split.", "aten.split_with_sizes.", "aten.unbind.", "aten.narrow.", "aten.is_same_size.",
             "aten.lift_fresh.", "aten._local_scalar_dense.", "aten.sym_", "prim.", "aten.size.", "aten.stride.", "aten.storage_offset.", "aten.numel.")


_NATIVE_STAGE = cfg.stage_native      # A/B switch: the input staging as one native pass


class _Proxy:
    """What lib.load() returns while a tape records: tape-able entry points are wrapped, everything else is the library's own function."""

    def __init__(self, real, tape):
        self.__dict__["_real"], self.__dict__["_tape"], self.__dict__["_cache"] = real, tape, {}

    def __getattr__(self, name):
        w = self._cache.get(name)
        if w is None:
            fn = getattr(self._real, name)
            fid = self._real.maf_tape_fn_id(name.encode())
            w = fn if fid == -1000 else self._tape._wrap(name, fid, fn)
            self._cache[name] = w
        return w


class StepTape:
    def __init__(self, ex, dev):
        self.ex, self.dev = ex, dev
        self.keep = []                           # every tensor / ctypes object a recorded call refers to
        self.lists = {"fwd": [], "bwd": []}      # MafTapeRec objects while recording
        self.toggles = {"fwd": [], "bwd": []}    # (record index or None, slot / ctypes array + index, mask, width)
        self.arr, self.n, self.tog = {}, {}, {}  # finalised: contiguous record arrays, counts, toggle tables
        self.marks = []                          # (backward record index, bucket index, bucket had main-stream contributions): where a bucket's all-reduce went out
        self.phase = None                        # "fwd" / "bwd" while a region records
        self.ready = False
        self.failed = None                       # why the recording was dropped (the eager path stays in use)
        self.glue = []                           # debug (`with tape.check():` around the recorded step): device ops torch ran inside the recorded regions
        self.seen = {"fwd": 0, "bwd": 0}         # ... and how many aten calls the check saw per region at all (zero = the mode did not reach that thread)
        self.pending_backward = False            # a replayed forward whose backward has not run: another forward must not overwrite the static buffers
        self._live = None                        # weak reference to that forward's autograd node: dead = its graph was dropped and no backward will ever come
        self.dropped = 0                         # replayed forwards whose backward never ran (a skipped step, an exception in the loss, a forward used for statistics only)
        self.xin = None
        self.outs, self.gin, self.gin_views = [], [], []
        self.stats_delta = {}
        self.main = self.side = None
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)

    # ------------------------------------------------------------------ recording
    def _wrap(self, name, fid, fn):
        tape, argtypes = self, fn.argtypes

        def wrapped(*args):
            rc = fn(*args)
            if rc == 0 and tape.phase is not None:
                tape._append(name, fid, argtypes, args)
            return rc
        return wrapped

    def _append(self, name, fid, argtypes, args):
        recs = self.lists[self.phase]
        r = lib.MafTapeRec()
        r.fn = fid
        if fid == -1:                            # maf_stream_fork(src, dst): the record holds the two stream indices
            r.a[0], r.a[1] = self._sidx(args[0], name), self._sidx(args[1], name)
            recs.append(r)
            return
        if len(args) != len(argtypes) or len(args) > lib.TAPE_MAX_ARGS:
            raise lib.MafError("step tape: %s called with %d arguments (declared: %d)" % (name, len(args), len(argtypes)))
        r.stream = self._sidx(args[-1], name)
        for i, (a, at) in enumerate(zip(args, argtypes)):
            r.a[i] = self._word(a, at, len(recs), i)
        recs.append(r)

    def _sidx(self, st, name):
        st = st.value if isinstance(st, C.c_void_p) else st
        i = self.sidx.get(st or 0)
        if i is None:
            raise lib.MafError("step tape: %s on a stream the recording does not know (main, weight-gradient stream, lanes)" % name)
        return i

    def _word(self, a, at, ri, slot):
        if isinstance(a, lib.Phase):
            self.toggles[self.phase].append((ri, slot, 1, 8))
            return int(a)
        if a is None:
            return 0
        if isinstance(a, bool):
            return int(a)
        if isinstance(a, int):
            return a & _M64
        if isinstance(a, float):
            if at is C.c_float:
                return struct.unpack("<I", struct.pack("<f", a))[0]
            if at is C.c_double:
                return struct.unpack("<Q", struct.pack("<d", a))[0]
            raise lib.MafError("step tape: a float for a non-float parameter")
        if isinstance(a, C.Array):
            self.keep.append(a)
            return C.addressof(a)
        obj = getattr(a, "_obj", None)          # C.byref(x)
        if obj is not None:
            if isinstance(obj, lib.MafOp):       # train_ops caches launch descriptors by geometry and rewrites their pointers per call: the tape keeps its own copy
                tog = getattr(obj, "_tape_toggles", None)
                obj = lib.MafOp.from_buffer_copy(obj)
                for off, mask, width in tog or ():                               # words of the descriptor that alternate between replays (a BatchNorm scratch half)
                    self.toggles[self.phase].append((None, C.addressof(obj) + off, mask, width))
            self.keep.append(obj)
            return C.addressof(obj)
        if isinstance(a, C._SimpleCData):
            return (a.value or 0) & _M64
        if isinstance(a, (bytes, bytearray)):
            self.keep.append(a)
            return C.cast(C.c_char_p(bytes(a)), C.c_void_p).value
        raise lib.MafError("step tape: cannot record an argument of type %s" % type(a).__name__)

    def toggle_array(self, arr, items):
        """Words of a host array a recorded call reads (kept alive here) that alternate between replays: items = [(index, xor mask, width in bytes)]."""
        if self.phase is None:
            return
        self.keep.append(arr)
        base, es = C.addressof(arr), C.sizeof(arr._type_)
        for j, mask, width in items:
            self.toggles[self.phase].append((None, base + j * es, mask, width))

    def begin(self, which):
        """Start recording region `which` ("fwd": from the step's pack batch to the head outputs; "bwd": from the boundary node to the end of the engine's pass)."""
        self._set_streams()
        self.phase = which
        lib._recorder = _Proxy(lib._lib, self)
        train_ops._keep, train_ops._rec = self.keep, self

    def _set_streams(self):
        """[main, weight-gradient stream, lanes ...] as raw handles: what the records' stream indices mean (the main stream is whatever is current now)."""
        self.main = train_ops._stream(self.dev) or 0
        self.side = (train_ops.side_stream(self.dev).cuda_stream or 0) if train_ops.wgrad_stream else self.main
        hs = [self.main, self.side] + [h or 0 for h in train_ops.lane_handles(self.dev)]
        self.sidx = {}
        for i, h in enumerate(hs):
            self.sidx.setdefault(h, i)
        self.harr = (C.c_void_p * len(hs))(*hs)

    def end(self):
        self.phase = None
        lib._recorder = None
        train_ops._keep = train_ops._rec = None

    def abort(self, why):
        self.end()
        self.failed = why
        self.ready = False


    def mark_bucket(self, index, main_contrib):
        if self.phase == "bwd":
            self.marks.append((len(self.lists["bwd"]), index, bool(main_contrib)))

    def _finalise(self, which):
        recs = self.lists[which]
        n = len(recs)
        arr = (lib.MafTapeRec * max(n, 1))()
        for i, r in enumerate(recs):
            C.memmove(C.addressof(arr) + i * _REC, C.addressof(r), _REC)
        tg = self.toggles[which]
        tab = (lib.MafTapeToggle * max(len(tg), 1))()
        for t, (ri, slot, mask, width) in zip(tab, tg):
            t.addr = (C.addressof(arr) + ri * _REC + _A_OFF + 8 * slot) if ri is not None else slot
            t.mask, t.width = mask & _M64, width
        self.arr[which], self.n[which], self.tog[which] = arr, n, (tab, len(tg))
        self.lists[which] = []
        self._toggle(which)                      # the recorded step has used phase 0 of every scratch: the first replay takes the other half

    def _toggle(self, which):
        tab, n = self.tog[which]
        if n:
            lib.check(lib._lib.maf_tape_toggle(tab, n))

    # ------------------------------------------------------------------ the recorded step
    def record_forward(self, model, x):
        """Run the train-form forward of `model` on image batch x the normal way, recording it — Detect's train-branch join (train_ops.detect_join) included.  Returns
        (stem feature maps per level, cls [B,A,nc], reg [B,A,4*(reg_max+1)]) behind the boundary node (the gradients of cls / reg are copied into static buffers before
        the recorded backward kernels read them)."""
        B, _, H, W = x.shape
        self.xin = torch.zeros((B, 8, H, W), dtype=torch.float16, device=self.dev).contiguous(memory_format=torch.channels_last)
        self._stage_input(x)
        stats0 = dict(train_ops.stats)
        ex0 = dict(self.ex.stats)
        self.begin("fwd")
        try:
            heads = model._forward_train_form(self.xin, raw_heads=True)
            cls, reg = train_ops.detect_join(heads)
        except BaseException:
            self.abort("the recorded forward raised")
            raise
        self.end()
        self._finalise("fwd")
        self.nlev = len(heads)
        flat = [h[0] for h in heads] + [cls, reg]                                  # the stem feature maps (shapes for the loss, no gradient), then the two joined tensors
        self.outs = [t.detach() for t in flat]
        self.gin = [None] * self.nlev + [torch.zeros_like(t) for t in (cls, reg)]  # contiguous [B,A,C] in the heads' dtype: what the recorded join-backward reads
        self.gin_views = list(self.gin)
        self._stats0, self._ex0 = stats0, ex0
        out = _Boundary.apply(self, *flat)
        return list(out[:self.nlev]), out[self.nlev], out[self.nlev + 1]

    def _begin_backward_record(self):
        self.begin("bwd")
        torch.autograd.Variable._execution_engine.queue_callback(self._end_backward_record)

    def _end_backward_record(self):
        if self.phase != "bwd":
            return
        self.end()
        self._finalise("bwd")
        self.stats_delta = {k: v - self._stats0.get(k, 0) for k, v in train_ops.stats.items() if v != self._stats0.get(k, 0)}
        off = {k: v for k, v in self.stats_delta.items() if k in ("fallback", "torch_bn", "torch_maxpool", "framework_wgrad_fp32", "glue", "conv_tuned")}
        if off:                                  # something of the step ran as a torch kernel (or timed conv variants): it would not be in the lists
            self.failed = "the recorded step left the HIP path: %s" % off
            return
        self.ex_delta = {k: v - self._ex0.get(k, 0) for k, v in self.ex.stats.items() if v != self._ex0.get(k, 0) and k != "collectives"}
        if self.glue:
            self.failed = "torch ran device ops inside the recorded regions: %s" % sorted(set(self.glue))
            return
        self.ready = True

    # ------------------------------------------------------------------ replay
    def _stage_input(self, x):
        """The image batch into the static NHWC8 fp16 input: one native pass for a contiguous fp32 / fp16 NCHW batch (torch's copy_ into the channel slice: four kernels, 186 us
        at batch 32), torch's copy otherwise."""
        if _NATIVE_STAGE and x.is_contiguous() and x.dtype in (torch.float32, torch.float16) and (x.shape[2] * x.shape[3]) % 4 == 0:
            B, _, H, W = x.shape
            L = lib._lib if lib._lib is not None else lib.load()                   # (never recorded: it runs in front of the lists, every step)
            lib.check(L.maf_image_to_nhwc8(x.data_ptr(), lib.F32 if x.dtype == torch.float32 else lib.F16, B, H, W, self.xin.data_ptr(), train_ops._stream(self.dev)))
        else:
            self.xin[:, :3].copy_(x)

    def _run(self, which, first, last):
        bad = C.c_int32(-1)
        rc = lib._lib.maf_tape_run(self.arr[which], first, last, self.harr, len(self.harr), C.byref(bad))
        if rc:
            raise lib.MafError("step tape: %s record %d failed: %s" % (which, bad.value, lib._lib.maf_last_error().decode()))

    def replay_forward(self, x):
        self._stage_input(x)
        self.ex.begin()
        out = _TapeStep.apply(self, self._anchor)
        for k, v in self.stats_delta.items():
            train_ops.stats[k] = train_ops.stats.get(k, 0) + v
        train_ops.stats["tape_replays"] = train_ops.stats.get("tape_replays", 0) + 1
        return list(out[:self.nlev]), out[self.nlev], out[self.nlev + 1]

    def release(self):
        """The model drops this tape: forget the buffers it pinned and their entries in train_ops.zero_padded (keyed by address: a later allocation at the same
        address must not inherit 'the channels behind are zero')."""
        for v in self.keep:                                                        # (the padded per-level gradient maps of the recorded join-backward are among these)
            if isinstance(v, torch.Tensor):
                train_ops.zero_padded.pop(v.data_ptr(), None)
        self.ready = False
        self.failed = self.failed or "released"
        self.keep, self.outs, self.gin, self.gin_views, self.arr, self.tog = [], [], [], [], {}, {}
        self.xin = None

    def graph_alive(self):
        """Is the autograd node of the last replayed forward still reachable (its outputs, or a loss computed from them, are held somewhere)?"""
        return self._live is not None and self._live() is not None

    def drop_pending(self):
        """The last replayed forward never got its backward and its graph is gone (a non-finite loss that skipped the step, an exception in the loss, a forward
        that only refreshed BatchNorm statistics): release the static buffers for the next forward.  The forward list has toggled the scratch halves of ITS call
        sites after running; the backward list did not run and keeps its phase — every recorded call site owns its scratch (train_ops._bn_part under a recording),
        a backward site accumulates into the half its own last run cleared, so its phase advances with its own runs only (toggling it here would point it at the
        half that still holds the last backward's sums: measured, gradients off by 4x).  (Without the release the tape stayed 'busy' for ever and every later
        step silently took the eager path: ADVICE round 5.)"""
        self.pending_backward = False
        self._live = None
        self.dropped += 1
        train_ops.stats["tape_dropped_backward"] = train_ops.stats.get("tape_dropped_backward", 0) + 1
        if self.dropped == 1:
            import warnings
            warnings.warn("maf_yolo_amd step tape: a replayed train-mode forward was not followed by its backward (skipped step?); its buffers were released for the "
                          "next forward.  Counted in train_ops.stats['tape_dropped_backward'].")

    def _copy_grads(self, grads):
        for i, g in enumerate(grads):
            if self.gin_views[i] is None:
                continue
            if g is None:
                self.gin_views[i].zero_()
            else:
                self.gin_views[i].copy_(g)

    def _replay_backward(self, grads):
        ex = self.ex
        self._copy_grads(grads)
        ex.ensure_attached()
        ex._arm()                                                                  # finish() closes the pass: flushes open buckets, joins the weight-gradient stream
        pos = 0
        for idx, bi, main in self.marks:
            self._run("bwd", pos, idx)
            pos = idx
            ex.replay_launch(bi, main)
        self._run("bwd", pos, self.n["bwd"])
        if train_ops.wgrad_stream:
            train_ops._side_used[self.dev.index] = True
        self._toggle("bwd")
        for k, v in self.ex_delta.items():
            ex.stats[k] += v
        self.pending_backward = False
        self._live = None


class _Boundary(torch.autograd.Function):
    """Identity on the head outputs of a RECORDED step; its backward puts the incoming gradients into the tape's static buffers (what the recorded backward
    kernels read) and opens the backward recording."""

    @staticmethod
    def forward(ctx, tape, *ts):
        ctx.tape = tape
        return tuple(t.view_as(t) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        tape = ctx.tape
        tape._copy_grads(gs)
        tape._begin_backward_record()
        return (None,) + tuple(None if v is None else v for v in tape.gin_views)


class _TapeStep(torch.autograd.Function):
    """A replayed step as ONE autograd node: forward = the recorded forward list, backward = the recorded backward list."""

    @staticmethod
    def forward(ctx, tape, anchor):
        ctx.tape = tape
        tape._run("fwd", 0, tape.n["fwd"])
        tape._toggle("fwd")
        tape.pending_backward = True
        tape._live = weakref.ref(ctx)
        outs = tuple(t.view_as(t) for t in tape.outs)
        ctx.mark_non_differentiable(*outs[:tape.nlev])
        return outs

    @staticmethod
    def backward(ctx, *gs):
        ctx.tape._replay_backward(gs)
        return None, None


class check(torch.utils._python_dispatch.TorchDispatchMode):
    """`with tape.check():` around whole training steps (forward, loss, backward — entered on the calling thread, the autograd engine carries the mode to its
    workers): while a step tape records, every aten call inside a recorded region that touches device tensors and is neither a view nor an allocation lands
    in that tape's `glue` list — a torch kernel between recorded launches, which a replay would not run.  A tape with a non-empty list refuses to replay."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        tape = train_ops._rec
        ph = None if tape is None else tape.phase
        if ph is None:
            return out
        tape.seen[ph] += 1
        name = str(func)
        if not name.startswith(_VIEW_OPS) and not train_ops._in_alloc:
            def dev(o):
                if isinstance(o, torch.Tensor):
                    return o.is_cuda
                if isinstance(o, (list, tuple)):
                    return any(dev(i) for i in o)
                return False
            if dev(args) or dev(out) or dev(list((kwargs or {}).values())):
                tape.glue.append(name)
                if len(tape.glue) <= 4:                                          # where it came from (the first few): the Python frames of the call
                    import traceback
                    tape.glue_where = getattr(tape, "glue_where", []) + ["".join(traceback.format_stack(limit=14)[:-1])]
        return out
