"""The gradient exchange of the data-parallel train step, owned by this package instead of borrowed from DistributedDataParallel.

Reference: `Trainer.train_in_steps` (yolov6/core/engine.py:141-167) runs backward under DDP (engine.py:477-489 wraps the model), whose
reducer copies every gradient into a bucket from its AccumulateGrad hook ON THE MAIN STREAM and launches one all-reduce per bucket.  Here
the weight gradients are produced on a SIDE stream (train_ops._fork: dW of a layer depends on nothing that follows it in the backward
chain), so a reducer that reads them on the main stream forces a join after every layer.  `GradExchange` keeps the whole chain

    weight-gradient kernel  ->  fold into the bucket  ->  all-reduce of the bucket

on the side stream:

* every parameter owns a slice of a flat fp32 bucket; `p.grad` is a VIEW of that slice (`zero_grad()` is one memset per bucket; an
  `optimizer.zero_grad()` with its default `set_to_none=True` — what the reference trainer calls, engine.py:347,388 — is survived: the first
  gradient of the next pass finds the view gone, clears the slice and re-attaches it, `_attach`);
* the conv weight-gradient kernels accumulate straight into the slice (1x1, depth-wise: the kernel's layout is the parameter's) or go
  through one `maf_grad_fold` launch (3x3: tap-major -> [Cout][Cin][3][3]; padded channel counts) — on the side stream, and the autograd
  Function returns None for the weight, so no AccumulateGrad node touches these gradients on the main stream;
* gradients that autograd produces itself (BatchNorm affine, biases, the 1x1 depth-wise scales, anything on CPU) arrive through
  `register_post_accumulate_grad_hook` — accumulated in place into the same views on the main stream;
* buckets are filled in reverse registration order (the order backward reaches the layers); when the last gradient of a bucket has been
  issued the bucket's all-reduce is launched from the side stream (`dist.all_reduce(async_op=True)` under a side-stream context: RCCL
  orders the collective behind what the side stream holds; a bucket with main-stream contributions first waits for an event recorded
  on the main stream), so the data-gradient chain on the main stream never waits for a weight gradient or a collective;
* the main stream waits ONCE, at the end of backward (`finish()`, queued as an autograd-engine callback by the first gradient of a
  pass): side stream + every outstanding collective; after that the GradScaler / optimizer may read the gradients.

World size 1 runs the SAME schedule minus the collectives (bench.py times N = 1 through it, so a 1 -> N scaling ratio compares like with
like).  `no_sync()` (gradient accumulation, engine.py:377-388) skips the collectives of a backward pass; the next synchronising pass
reduces the accumulated sum, like DDP."""
import contextlib

import torch

from .config import cfg
from . import lib

_DEBUG = cfg.exchange_debug    # keep the Python stack of every arrival (shown when a late arrival raises)
current = None                                  # the exchange train_ops hands its weight gradients to (None: plain autograd)


class _Bucket:
    __slots__ = ("flat", "params", "ids", "arrived", "main_contrib", "work", "launched")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.ids = frozenset(id(p) for p in params)
        self.arrived, self.main_contrib, self.work, self.launched = set(), False, None, False

    def reset(self):
        self.arrived, self.main_contrib, self.work, self.launched = set(), False, None, False


class GradExchange:
    def __init__(self, model, bucket_bytes=None, process_group=None, world_size=None, force_collectives=False):
        """`force_collectives`: issue the bucket all-reduces even when the group has ONE rank (the RCCL schedule — AVG, async, from the side
        stream — then runs on a single GPU: tests/test_gpu_exchange.py, `bench.py --train --rccl1`)."""
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        self.world = world_size if world_size is not None else (self.dist.get_world_size(process_group) if self.dist is not None else 1)
        self.force = bool(force_collectives)
        if self.force and self.dist is None:
            raise lib.MafError("GradExchange(force_collectives=True) needs an initialised process group")
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "GradExchange: the model has no trainable parameter"
        self.names = {id(p): n for n, p in model.named_parameters()}
        self._via = {}                                                            # id(p) -> how its gradients of this pass arrived ("side" / "main")
        self.device = params[0].device
        total = sum(p.numel() for p in params) * 4
        if bucket_bytes is None:
            # ~6 collectives per step: each large enough to run at link bandwidth on a ring over xGMI (per-link bound, ~100 us of latency per
            # collective at 8 ranks), early enough that all but the last overlap the remaining backward
            bucket_bytes = max(4 << 20, -(-total // 6))
        self.buckets, self.slot = [], {}
        cur, size = [], 0
        for p in reversed(params):                                                # backward reaches the layers in (about) this order
            if p.dtype != torch.float32:
                raise lib.MafError("GradExchange: fp32 master parameters expected, got %s" % p.dtype)
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self._sync = True
        self._armed = False
        self._next = 0                                                           # buckets go out in bucket order on every rank (see _arrived)
        self._hooks = [p.register_post_accumulate_grad_hook(self._main_hook) for p in params]
        self.stats = {"collectives": 0, "side_direct": 0, "side_folded": 0, "main_hook": 0, "main_direct": 0, "reattached": 0}
        self.begin()                                                             # active from here on (`close()` deactivates)

    def _close(self, ps):
        n = sum(-(-p.numel() // 64) * 64 for p in ps)                            # slices start on 256-byte boundaries
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        b = _Bucket(flat, ps)
        off = 0
        for p in ps:
            view = flat[off:off + p.numel()].view_as(p)
            p.grad = view
            self.slot[id(p)] = (b, view)
            off += -(-p.numel() // 64) * 64
        self.buckets.append(b)

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """One memset per bucket; re-attaches `p.grad` views an optimizer's zero_grad(set_to_none=True) may have dropped."""
        for b in self.buckets:
            b.flat.zero_()
        for b in self.buckets:
            for p in b.params:
                if p.grad is None or p.grad.data_ptr() != self.slot[id(p)][1].data_ptr():
                    p.grad = self.slot[id(p)][1]

    def _reset_pass(self):
        self._armed = False
        self._next = 0
        self._via = {}
        for b in self.buckets:
            b.reset()

    def begin(self):
        """Start of a forward: this exchange receives the weight gradients of the next backward pass.  Resets the per-pass state, so a
        backward pass that raised cannot leave a stale 'callback queued' flag behind (`finish()` resets it too: a second backward without
        a new forward — retain_graph, two forwards then two backwards — starts clean)."""
        global current
        current = self
        self._reset_pass()

    @contextlib.contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    # ------------------------------------------------------------------ gradient arrival
    def _arm(self):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)

    def _attach(self, p, view):
        """`p.grad` must BE the bucket view.  The reference trainer calls `optimizer.zero_grad()` (yolov6/core/engine.py:347,388), whose default
        `set_to_none=True` drops it: the slice then still holds the previous step's averaged gradient, which 'gradients were cleared' means
        to be zero.  Runs on the main stream BEFORE the weight-gradient kernel is forked to the side stream (the fork's event orders it)."""
        g = p.grad
        if g is view or (g is not None and g.data_ptr() == view.data_ptr()):
            return
        if g is None:
            view.zero_()
        else:                                                                     # a foreign tensor (someone assigned p.grad): it IS the accumulated gradient
            view.copy_(g)
        p.grad = view
        self.stats["reattached"] += 1

    def target(self, p):
        """(bucket, view) of parameter p if it is registered (train_ops: where the weight gradient goes), else None.  Called once per
        weight-gradient launch, before the fork to the side stream."""
        ent = self.slot.get(id(p))
        if ent is not None:
            self._attach(p, ent[1])
        return ent

    def side_done(self, p, folded=False):
        """train_ops: the gradient of p has been issued into its slice on the side stream."""
        self._arm()
        b = self.slot[id(p)][0]
        self.stats["side_folded" if folded else "side_direct"] += 1
        self._arrived(b, p, False)

    def main_done(self, p):
        """train_ops: a kernel on the MAIN stream has added the gradient of p to its slice (the BatchNorm affine parameters: csrc/bn_act.hip adds
        dgamma / dbeta in its apply pass; the Function returns None for them, so no AccumulateGrad add kernel — 280 per step of MAF-YOLO-n)."""
        self._arm()
        self.stats["main_direct"] += 1
        self._arrived(self.slot[id(p)][0], p, True, "direct")

    def _main_hook(self, p):
        ent = self.slot.get(id(p))
        if ent is None:
            return
        b, view = ent
        via = self._via.get(id(p), ())
        if "side" in via or "direct" in via:
            # torch calls the post-accumulate hook of a leaf whose Function returned None as well (measured on torch 2.10: all 124 conv weights of
            # MAF-YOLO-n, tensor hooks see `None`): the gradient of this pass went into the slice from a kernel (side stream: conv weights; main
            # stream: BatchNorm affine) and was counted there.
            # (Round 3 counted both calls: buckets "completed" early — harmless at world size 1, wrong for N > 1; found by the one-rank RCCL test.)
            return
        if p.grad is not view and (p.grad is None or p.grad.data_ptr() != view.data_ptr()):
            # autograd found `p.grad` empty (zero_grad(set_to_none=True) before this pass) and installed a fresh tensor: that tensor is the
            # whole gradient — the slice's old content is the step before's — so it REPLACES the slice
            view.copy_(p.grad)
            p.grad = view
            self.stats["reattached"] += 1
        self._arm()
        self.stats["main_hook"] += 1
        self._arrived(b, p, True)

    def _arrived(self, b, p, on_main, tag=None):
        if b.launched and self._sync and self._collectives():
            # the bucket has gone out with this pass's first contribution of every parameter; one more would be lost on the other ranks
            raise lib.MafError("GradExchange: a gradient of %s arrived (%s stream) after its bucket was reduced; earlier arrivals of this pass: %s "
                               "(a parameter used more than once per backward pass is not supported)"
                               % (self.names.get(id(p), "?"), "main" if on_main else "side", self._via.get(id(p))))
        b.main_contrib = b.main_contrib or on_main
        b.arrived.add(id(p))
        self._via.setdefault(id(p), []).append(tag or ("main" if on_main else "side"))
        if _DEBUG:
            import traceback
            self._via[id(p)].append("".join(traceback.format_stack(limit=8)))
        if not self._sync:
            return
        # in BUCKET ORDER on every rank: a parameter without a gradient on one rank only must not reorder the collectives between ranks
        # (RCCL matches them by issue order) — a complete bucket waits for the ones in front of it, `finish()` flushes the rest in order
        while self._next < len(self.buckets) and len(self.buckets[self._next].arrived) == len(self.buckets[self._next].ids):
            self._launch(self.buckets[self._next])
            self._next += 1

    def _collectives(self):
        return self.dist is not None and (self.world > 1 or self.force)

    def _launch(self, b):
        if b.launched:
            return
        b.launched = True
        from . import train_ops
        if train_ops._rec is not None:                                           # a recording step tape (tape.py) cuts its backward list here: a replay launches the bucket from the same place
            train_ops._rec.mark_bucket(self.buckets.index(b), b.main_contrib)
        if not self._collectives():
            return
        dist = self.dist
        avg = dist.ReduceOp.AVG if dist.get_backend(self.group) == "nccl" else dist.ReduceOp.SUM
        if b.flat.is_cuda:
            from . import train_ops
            side = train_ops.side_stream(b.flat.device)
            if b.main_contrib:                                                    # gradients autograd accumulated on the main stream
                train_ops.join_lanes(b.flat.device)                               # ... or kernels added on a lane of a step tape (BatchNorm affine gradients of the head branches)
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
            with torch.cuda.stream(side):
                b.work = dist.all_reduce(b.flat, op=avg, group=self.group, async_op=True)
                if avg == dist.ReduceOp.SUM:
                    b.work.wait()
                    b.flat.mul_(1.0 / self.world)
                    b.work = None
        else:
            b.work = dist.all_reduce(b.flat, op=avg, group=self.group, async_op=True)
        self.stats["collectives"] += 1

    # ------------------------------------------------------------------ replayed passes (tape.py)
    def ensure_attached(self):
        """Before a replayed backward: its kernels add into the bucket slices by address, so `p.grad` must be the views — an `optimizer.zero_grad()` with
        set_to_none=True (the reference trainer's, engine.py:347,388) has dropped them and means 'the gradients are zero'."""
        # per parameter, with the eager path's rule (`_attach`): None -> the slice is zeroed; a foreign tensor (assigned by the caller, restored from a checkpoint) IS
        # the accumulated gradient -> copied into the slice; then `p.grad` is the view.  A pure Python loop with no device work when everything is attached.
        n0 = self.stats["reattached"]
        for b in self.buckets:
            if all(p.grad is None for p in b.params):                             # the usual case after zero_grad(set_to_none=True): one memset for the bucket
                b.flat.zero_()
                for p in b.params:
                    p.grad = self.slot[id(p)][1]
                self.stats["reattached"] += 1
                continue
            for p in b.params:
                self._attach(p, self.slot[id(p)][1])
        return self.stats["reattached"] - n0

    def replay_launch(self, index, main_contrib):
        """A replayed backward has issued every gradient of bucket `index` (the point the recording marked): its all-reduce goes out, as in `_arrived`."""
        if not self._sync:
            return
        b = self.buckets[index]
        b.main_contrib = bool(main_contrib)
        self._launch(b)
        self._next = max(self._next, index + 1)

    def finish(self):
        """End of backward: launch what is still open, in bucket order (parameters that received no gradient leave their bucket incomplete),
        then make the main stream wait for the side stream and for every collective, and reset the per-pass state."""
        try:
            if self._sync:
                late = self.buckets[self._next:]
                if late and any(b_.arrived for b_ in self.buckets):
                    # buckets that never completed during backward (a parameter without a gradient in this pass holds its bucket AND every later one: they
                    # go out in bucket order) are reduced only here, with no backward left to hide them behind: counted, and named once
                    self.stats["late_buckets"] = self.stats.get("late_buckets", 0) + len(late)
                    if not getattr(self, "_late_warned", False):
                        self._late_warned = True
                        missing = [self.names.get(id(p_), "?") for b_ in late for p_ in b_.params if id(p_) not in b_.arrived][:8]
                        import warnings
                        warnings.warn("GradExchange: %d of %d buckets were reduced only at the end of backward (no overlap); parameters without a gradient in this pass: %s"
                                      % (len(late), len(self.buckets), missing))
                for b in late:
                    b.main_contrib = True
                    self._launch(b)
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
                    if not b.flat.is_cuda and self.dist is not None and self.dist.get_backend(self.group) != "nccl":
                        b.flat.mul_(1.0 / self.world)
                    b.work = None
            if self.device.type == "cuda":
                from . import train_ops
                train_ops.join_side(self.device)
        finally:
            self._reset_pass()

    def close(self):
        global current
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if current is self:
            current = None
