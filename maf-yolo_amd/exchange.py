"""The gradient exchange of the data-parallel train step, owned by this package instead of borrowed from DistributedDataParallel.

Reference: `Trainer.train_in_steps` (yolov6/core/engine.py:141-167) runs backward under DDP (engine.py:477-489 wraps the model), whose
reducer copies every gradient into a bucket from its AccumulateGrad hook ON THE MAIN STREAM and launches one all-reduce per bucket.  Here
the weight gradients are produced on a SIDE stream (train_ops._fork: dW of a layer depends on nothing that follows it in the backward
chain), so a reducer that reads them on the main stream forces a join after every layer.  `GradExchange` keeps the whole chain

    weight-gradient kernel  ->  fold into the bucket  ->  all-reduce of the bucket

on the side stream:

* every parameter owns a slice of a flat fp32 bucket; `p.grad` is a VIEW of that slice (set once; `zero_grad()` is one memset per
  bucket, never `set_to_none`);
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

from . import lib

current = None                                  # the exchange train_ops hands its weight gradients to (None: plain autograd)


class _Bucket:
    __slots__ = ("flat", "params", "pending", "main_contrib", "work", "launched")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending, self.main_contrib, self.work, self.launched = 0, False, None, False


class GradExchange:
    def __init__(self, model, bucket_bytes=None, process_group=None, world_size=None):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        self.world = world_size if world_size is not None else (self.dist.get_world_size(process_group) if self.dist is not None else 1)
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "GradExchange: the model has no trainable parameter"
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
        self._hooks = [p.register_post_accumulate_grad_hook(self._main_hook) for p in params]
        self.stats = {"collectives": 0, "side_direct": 0, "side_folded": 0, "main_hook": 0}
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

    def begin(self):
        """Start of a forward: this exchange receives the weight gradients of the next backward pass.  Resets the per-pass state, so a
        backward pass that raised cannot leave a stale 'callback queued' flag behind."""
        global current
        current = self
        self._armed = False
        for b in self.buckets:
            b.pending, b.main_contrib, b.work, b.launched = len(b.params), False, None, False

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

    def target(self, p):
        """(bucket, view) of parameter p if it is registered (train_ops: where the weight gradient goes), else None."""
        return self.slot.get(id(p))

    def side_done(self, p, folded=False):
        """train_ops: the gradient of p has been issued into its slice on the side stream."""
        self._arm()
        b = self.slot[id(p)][0]
        self.stats["side_folded" if folded else "side_direct"] += 1
        self._arrived(b, False)

    def _main_hook(self, p):
        ent = self.slot.get(id(p))
        if ent is None:
            return
        b, view = ent
        if p.grad is not view and (p.grad is None or p.grad.data_ptr() != view.data_ptr()):
            # someone replaced the view (zero_grad(set_to_none=True) before this pass): fold the fresh tensor back into the bucket
            view.add_(p.grad)
            p.grad = view
        self._arm()
        self.stats["main_hook"] += 1
        self._arrived(b, True)

    def _arrived(self, b, on_main):
        b.main_contrib = b.main_contrib or on_main
        b.pending -= 1
        if b.pending == 0 and self._sync:
            self._launch(b)

    def _launch(self, b):
        if b.launched or self.world <= 1 or self.dist is None:
            b.launched = True
            return
        b.launched = True
        dist = self.dist
        avg = dist.ReduceOp.AVG if dist.get_backend(self.group) == "nccl" else dist.ReduceOp.SUM
        if b.flat.is_cuda:
            from . import train_ops
            side = train_ops.side_stream(b.flat.device)
            if b.main_contrib:                                                    # gradients autograd accumulated on the main stream
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

    def finish(self):
        """End of backward: launch what is still open (parameters that received no gradient leave their bucket incomplete), then make the
        main stream wait for the side stream and for every collective."""
        if self._sync:
            for b in self.buckets:
                if not b.launched:
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
        self._armed = False

    def close(self):
        global current
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if current is self:
            current = None
