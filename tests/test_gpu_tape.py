"""-m gpu: the step tape (maf_yolo_amd/tape.py) — the recorded launch lists of a train step replayed by one C call each — against the eager step it records.

Reference step: yolov6/core/engine.py:141-167 (forward under autocast, ComputeLoss, scaled backward).  The eager path (layers.py -> train_ops.py, one autograd
Function per op) is what tests/test_gpu_train.py pins to the reference's gradients; these tests pin the replayed step to the eager one: same kernels, same order,
same buffers — what differs is who issues them."""
import pytest
import torch

import maf_yolo_amd as M
from maf_yolo_amd import exchange, synth, tape as tape_mod, train_ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _model(scale="n", step_tape="auto"):
    m = M.Model(scale)
    m.load_state_dict(synth.synth_state_dict(m, scale, 0))
    m = m.to(DEV).train()
    m.step_tape = step_tape
    return m


def _batch(bs, size, seed=1):
    x = synth.synth_images(bs, size, seed=seed).to(DEV)
    g = torch.Generator().manual_seed(100 + seed)
    nbox = 5 * bs
    wh = torch.rand(nbox, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(nbox, 2, generator=g) * (1 - wh)
    t = torch.cat([torch.arange(bs).repeat_interleave(5)[:, None].float(), torch.randint(0, 80, (nbox, 1), generator=g).float(), ctr, wh], 1).to(DEV)
    return x, t


def _pass(model, ex, crit, x, t, scale=1024.0):
    """forward + loss + backward of one batch into the exchange's (cleared) buckets; returns the loss"""
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    loss = crit((feats, cls, reg), t, 0, 0)[0]
    ex.zero_grad()
    (loss * scale).backward()
    return float(loss.detach())


def _grads(model):
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}


def _dev(got, ref, tol=1e-2):
    """worst parameter of `got` against `ref`: max |d| over (tol * the parameter's max |g| + 1e-4 of the model's largest gradient)"""
    G = max(float(v.abs().max()) for v in ref.values())
    return max((float((got[n] - ref[n]).abs().max()) / (tol * float(ref[n].abs().max()) + 1e-4 * G), n) for n in ref)


def _the_tape(model):
    (ent,) = model._tapes.values()
    return ent[1]


@pytest.mark.parametrize("scale", ["n", "s", "m"])
def test_tape_records_a_step_without_torch_kernels_between_its_launches(scale):
    """Steps 1-2 run eagerly (conv variants timed, weight staging plan filled), step 3 is recorded under the dispatch-mode check: no device op of torch's inside the
    recorded regions (it would be missing from a replay), the check did see both regions (also on the autograd engine's thread), the lists hold the step's launches."""
    model = _model(scale)
    ex = M.GradExchange(model)
    crit = M.ComputeLoss(ori_img_size=128, warmup_epoch=0)
    x, t = _batch(4, 128)
    try:
        with tape_mod.check():
            for _ in range(tape_mod.RECORD_AT):
                _pass(model, ex, crit, x, t)
        tp = _the_tape(model)
        assert tp is not None and tp.failed is None, (tp and tp.failed, tp and getattr(tp, "glue_where", None))
        assert tp.glue == [], sorted(set(tp.glue))
        assert tp.seen["fwd"] > 0 and tp.seen["bwd"] > 0, tp.seen
        assert tp.ready
        assert tp.n["fwd"] > 150 and tp.n["bwd"] > 300, tp.n       # C-ABI calls (a BatchNorm call is two kernels)
        assert len(tp.marks) == len(ex.buckets), (tp.marks, len(ex.buckets))
        before = train_ops.stats.get("tape_replays", 0)
        loss = _pass(model, ex, crit, x, t)
        assert train_ops.stats.get("tape_replays", 0) == before + 1 and loss == loss
    finally:
        ex.close()


@pytest.mark.parametrize("bs,size", [(4, 128), (32, 640)])
def test_replayed_step_gives_the_eager_steps_outputs_and_gradients(bs, size):
    """Three models with the same weights on the same batch: A and B step eagerly, C replays its tape.  The head outputs of the replayed forward equal the eager ones
    to the run-to-run noise of the BatchNorm atomics, and C's gradients are as close to A's as B's are (3 x the eager-vs-eager deviation, floor 0.5: both in units of
    1e-2 of a parameter's max |g|) — a stale buffer, a missing launch or a wrong phase shows as errors of order one."""
    crit = M.ComputeLoss(ori_img_size=size, warmup_epoch=0)
    x, t = _batch(bs, size)
    res = {}
    for name, mode in (("A", False), ("B", False), ("C", "auto")):
        model = _model(step_tape=mode)
        ex = M.GradExchange(model)
        try:
            n = tape_mod.RECORD_AT + 1 if mode else 1
            for _ in range(n):
                loss = _pass(model, ex, crit, x, t)
            torch.cuda.synchronize()
            if mode:
                tp = _the_tape(model)
                assert tp is not None and tp.ready and tp.failed is None, tp and tp.failed
            res[name] = (loss, _grads(model))
        finally:
            ex.close()
    la, ga = res["A"]
    lb, gb = res["B"]
    lc, gc = res["C"]
    assert abs(lc - la) <= max(3 * abs(lb - la), 2e-3 * abs(la)), (la, lb, lc)
    noise, tape_dev = _dev(gb, ga), _dev(gc, ga)
    assert tape_dev[0] <= max(3 * noise[0], 0.5), (tape_dev, noise)


def test_replayed_steps_follow_the_eager_training_trajectory():
    """Eight optimizer steps (GradScaler, fused SGD, EMA) eagerly and through the tape from the same weights: finite, and the losses stay together."""
    x, t = _batch(8, 128)
    crit = M.ComputeLoss(ori_img_size=128, warmup_epoch=0)
    out = {}
    for mode in (False, "auto"):
        model = _model(step_tape=mode)
        ex = M.GradExchange(model)
        opt = M.build_optimizer(model, lr0=0.002, momentum=0.9, weight_decay=5e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        ema = M.ModelEMA(model)
        losses = []
        try:
            for _ in range(8):
                with torch.autocast("cuda", dtype=torch.float16):
                    (feats, cls, reg), _ = model(x)
                loss = crit((feats, cls, reg), t, 0, 0)[0]
                ex.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
                ema.update(model)
                losses.append(float(loss.detach()))
            if mode:
                tp = _the_tape(model)
                assert tp is not None and tp.ready, tp and tp.failed
                assert train_ops.stats.get("tape_replays", 0) >= 8 - tape_mod.RECORD_AT
        finally:
            ex.close()
        out[mode] = losses
    a, b = out[False], out["auto"]
    assert all(v == v and abs(v) < 1e6 for v in a + b), (a, b)
    assert abs(a[-1] - b[-1]) <= 0.05 * abs(a[-1]), (a, b)
    assert a[-1] < a[0] and b[-1] < b[0], (a, b)


def test_second_forward_before_the_backward_runs_eagerly():
    """Two forwards, then two backwards: the second forward must not overwrite the static buffers the first one's backward will read — it takes the eager path."""
    model = _model()
    ex = M.GradExchange(model)
    crit = M.ComputeLoss(ori_img_size=128, warmup_epoch=0)
    x, t = _batch(4, 128)
    try:
        for _ in range(tape_mod.RECORD_AT):
            _pass(model, ex, crit, x, t)
        tp = _the_tape(model)
        assert tp.ready
        r0 = train_ops.stats.get("tape_replays", 0)
        with torch.autocast("cuda", dtype=torch.float16):
            (f1, c1, g1), _ = model(x)
            (f2, c2, g2), _ = model(x)
        assert train_ops.stats.get("tape_replays", 0) == r0 + 1
        l1 = crit((f1, c1, g1), t, 0, 0)[0]
        l2 = crit((f2, c2, g2), t, 0, 0)[0]
        ex.zero_grad()
        l2.backward()
        l1.backward()
        torch.cuda.synchronize()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        assert not tp.pending_backward
    finally:
        ex.close()


def test_forward_without_a_backward_releases_the_tape_and_the_next_step_is_correct():
    """A replayed forward whose backward never comes (a skipped step, an exception in the loss, a forward that only refreshed BatchNorm statistics) must not park the
    tape for ever: once its graph is gone the next forward replays again — counted, warned once — and the step after it gives the gradients of a
    model that never dropped a backward (the backward list's scratch phases advance with its own runs only)."""
    import warnings
    crit = M.ComputeLoss(ori_img_size=128, warmup_epoch=0)
    x, t = _batch(4, 128)
    res = {}
    for name, drop in (("A", False), ("B", False), ("C", True)):
        model = _model()
        ex = M.GradExchange(model)
        try:
            for _ in range(tape_mod.RECORD_AT):
                _pass(model, ex, crit, x, t)
            tp = _the_tape(model)
            assert tp.ready
            if drop:
                r0, d0 = train_ops.stats.get("tape_replays", 0), train_ops.stats.get("tape_dropped_backward", 0)
                with torch.autocast("cuda", dtype=torch.float16):
                    out = model(x)                                            # replayed forward ...
                assert tp.pending_backward and tp.graph_alive()
                del out                                                       # ... whose graph is dropped without a backward
                assert tp.pending_backward and not tp.graph_alive()
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    loss = _pass(model, ex, crit, x, t)                       # the next step replays again
                assert train_ops.stats.get("tape_replays", 0) == r0 + 2 and train_ops.stats.get("tape_dropped_backward", 0) == d0 + 1 and tp.dropped == 1
                assert any("not followed by its backward" in str(w_.message) for w_ in w)
                assert not tp.pending_backward
            else:
                loss = _pass(model, ex, crit, x, t)
            torch.cuda.synchronize()
            res[name] = (loss, _grads(model))
        finally:
            ex.close()
    (la, ga), (lb, gb), (lc, gc) = res["A"], res["B"], res["C"]
    # C ran one more train-mode forward (running statistics moved once more; the batch statistics the step normalises with are the same): same bars as eager-vs-replay
    assert abs(lc - la) <= max(3 * abs(lb - la), 2e-3 * abs(la)), (la, lb, lc)
    noise, dev = _dev(gb, ga), _dev(gc, ga)
    assert dev[0] <= max(3 * noise[0], 0.5), (dev, noise)


def test_replayed_backward_takes_over_a_foreign_gradient_tensor():
    """`p.grad` assigned by the caller (a restored checkpoint, gradient surgery) before a replayed backward IS the accumulated gradient: ensure_attached copies it into the
    bucket slice and re-attaches the view — parameter by parameter, as the eager `_attach` does — so the kernels' gradient adds to it and the optimizer steps on the bucket."""
    crit = M.ComputeLoss(ori_img_size=128, warmup_epoch=0)
    x, t = _batch(4, 128)
    model = _model()
    ex = M.GradExchange(model)
    try:
        for _ in range(tape_mod.RECORD_AT + 1):
            _pass(model, ex, crit, x, t)
        assert _the_tape(model).ready
        ref = _grads(model)
        names = [n for n, p in model.named_parameters() if p.requires_grad and p.dim() == 4][:2]
        ps = dict(model.named_parameters())
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = model(x)
        loss = crit((feats, cls, reg), t, 0, 0)[0]
        ex.zero_grad()
        ps[names[0]].grad = torch.full_like(ps[names[0]], 3.0)                 # a foreign tensor: the accumulated gradient so far
        ps[names[1]].grad = None                                              # dropped: means zero
        (loss * 1024.0).backward()
        torch.cuda.synchronize()
        for n in names:
            view = ex.slot[id(ps[n])][1]
            assert ps[n].grad.data_ptr() == view.data_ptr(), n
        # (two steps of this small batch differ by the run-to-run noise of the fp16 step — atomics, arg-max flips: up to 10 % of max |g| on single elements — so the
        #  check is on norms: a slice that kept the step before's gradient gives a relative error of 1.0, a lost foreign tensor shifts the median by 3)
        rel = lambda a, r: float((a - r).norm() / r.norm())
        g0, g1 = ps[names[0]].grad, ps[names[1]].grad
        assert rel(g0 - 3.0, ref[names[0]]) < 0.4 and abs(float((g0 - ref[names[0]]).median()) - 3.0) < 0.5, (rel(g0 - 3.0, ref[names[0]]), float((g0 - ref[names[0]]).median()))
        assert rel(g1, ref[names[1]]) < 0.4, rel(g1, ref[names[1]])
    finally:
        ex.close()
