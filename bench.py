#!/usr/bin/env python3
"""bench.py — images/s of the MAF-YOLO-n hot path on MI355X (BASELINE.json metric, configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 with no launcher around it re-executes itself as the line below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
Model.forward (deploy-form graph on the HIP engine, fp16 storage / fp32 accumulate) followed by
non_max_suppression (conf 0.03, IoU 0.65, multi_label — the settings of tools/eval.py) on
32 x 3 x 640 x 640 images per GPU.  With N > 1 every rank runs an independent replica on its own
batch (inference shards by image, no data-path collective => "weak" scaling);
the timed region is bracketed by a barrier + device sync on both sides and the MAX over ranks is
used.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      for the kernel (template instantiation) with the largest share of forward time:
                algorithmic bytes per launch (Plan.algorithmic_bytes: inputs once + outputs once +
                weights once, the SURVEY.md §8d layer-granular model) / its mean launch duration,
                measured here with HIP event pairs around every launch on the engine's stream.
  cpu_baseline  oracle (CPU restatement of the reference, fp32) on the host cores, bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The serving loop keeps three batches in flight on three streams plus one for the NMS.  ROCm multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run one after the other: give them room.  (Must be set
# before the HIP runtime starts, i.e. before torch is imported; an explicit setting in the environment wins.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this pool: the host driver supports dmabuf IPC only (without this RCCL fails with hipIpcGetMemHandle: invalid argument); already exported on the
# boxes, set here too so that a bare `python bench.py --gpus N` cannot miss it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense


def calibrate_cls_bias(model, x, target_per_img, M, torch):
    """Shift the cls_pred biases so that ~target_per_img (box,class) candidates per image exceed conf 0.03.
    The synthetic head would otherwise pass 5-30 % of all 672k pairs; a trained detector passes ~0.3 %."""
    with torch.no_grad():
        pred = model(x)[0]
        p = pred[..., 5:].float().clamp(1e-7, 1 - 1e-7)
        logit = torch.log(p) - torch.log1p(-p)
        k = max(1, int(target_per_img * pred.shape[0]))
        kth = torch.topk(logit.flatten(), k).values[-1].item()
        shift = float(torch.log(torch.tensor(0.03 / 0.97))) - kth
        for m in model.backbone:
            if hasattr(m, "cls_pred"):
                m.cls_pred.bias.add_(shift)
        model._plans = {}
    return shift


def latency_leg(torch, M, dev, scale, n, warmup, lanes=-1):
    """BASELINE configs[4]: bs=1, 640x640 fp16 — the forward issued eagerly from the C engine loop and replayed from a captured hipGraph (fixed input / output
    buffers), + NMS; per-call latency of n requests each.  Returns the result line as a dict."""
    import numpy as np
    from maf_yolo_amd import synth
    model = M.Model(scale)
    model.load_state_dict(synth.synth_state_dict(model, scale, 0))
    model = model.to(dev).eval()
    model.autotune = True                                  # the bs = 1 plan gets its own tile / variant choices, timed when the plan is built — outside every timed request,
    if lanes >= 0:                                         # like the headline's (until round 6 this leg ran the UNTUNED plan: 1.87 ms per forward where the tuned one takes 1.18)
        model.multi_stream = lanes
    x = synth.synth_images(1, 640, seed=1).to(dev).half()
    calibrate_cls_bias(model, x, 2000, M, torch)
    plan = model.plan_for(x)
    pred = torch.empty(1, plan.A, 5 + plan.nc, dtype=torch.float32, device=dev)
    res = {}
    for tag, graph in (("hipgraph", True), ("eager", False)):
        for _ in range(warmup):
            plan.run_into(x, pred, graph=graph)
            M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
        torch.cuda.synchronize(dev)
        lat_f, lat_t = [], []
        for _ in range(n):                                   # the forward alone (device idle before and after)
            t0 = time.perf_counter()
            plan.run_into(x, pred, graph=graph)
            torch.cuda.synchronize(dev)
            lat_f.append(1e3 * (time.perf_counter() - t0))
        for _ in range(n):                                   # the request: forward and NMS queued back to back on one stream, ONE host wait (the detections' counts)
            t0 = time.perf_counter()
            plan.run_into(x, pred, graph=graph)
            M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
            lat_t.append(1e3 * (time.perf_counter() - t0))
        res[tag] = {"forward_ms_p50": round(float(np.percentile(lat_f, 50)), 4), "forward_ms_p99": round(float(np.percentile(lat_f, 99)), 4),
                    "forward_plus_nms_ms_p50": round(float(np.percentile(lat_t, 50)), 4), "forward_plus_nms_ms_p99": round(float(np.percentile(lat_t, 99)), 4)}
    # BASELINE configs[4] names hipGraph capture as the mechanism; both ways of issuing the same launch list are measured and the line's value is the
    # faster one, named in the metric (a graph node costs ~1.1 us more than an eager launch from the C loop on this stack: DESIGN.md section 8)
    best = min(res, key=lambda k: res[k]["forward_plus_nms_ms_p50"])
    return {"metric": "latency ms MAF-YOLO-%s 640x640 bs=1 infer (forward + NMS, p50; launch path: %s)" % (scale, "hipGraph replay" if best == "hipgraph" else "eager launches from the C engine loop"),
            "value": res[best]["forward_plus_nms_ms_p50"], "unit": "ms", "n_gpus": 1, "higher_is_better": False,
            "dtype": "f16", "data": "synthetic", "config": {"workload": "bs=1 3x640x640 fp16, %d launches per forward" % len(plan.ops), "launch_path": best,
                                                            "samples": n, "warmup": warmup},
            "latency": res}


def latency_mode(args, torch, M, dev):
    print(json.dumps(latency_leg(torch, M, dev, args.scale, max(args.steps, 200), args.warmup, args.lanes)), flush=True)


def infer_leg(torch, M, dev, scale, batch, steps, warmup, cs=None):
    """The headline workload (forward + NMS of resident 640 x 640 fp16 batches, three batches in flight) for another scale, compact: the same serving loop as the
    main line's timed region with fewer steps, plus the forward alone.  Tile choices: the frozen file of the scale under profiles/ when there is one."""
    import numpy as np
    from maf_yolo_amd import synth, engine as _engine
    frozen = [p_ for p_ in (os.path.join(ROOT, "profiles", f_ % scale) for f_ in ("round6_tune_%s.json", "round5_tune_%s.json", "round4_tune_%s.json", "round3_tune_%s.json")) if os.path.exists(p_)]
    if frozen:
        _engine.load_tune_cache(frozen[0])
    model = M.Model(scale)
    model.load_state_dict(synth.synth_state_dict(model, scale, 0))
    model = model.to(dev).eval()
    model.autotune = True
    S = 3
    xs = [synth.synth_images(batch, 640, seed=1 + 7 * k).to(dev).half() for k in range(S)]
    calibrate_cls_bias(model, xs[0], 2000, M, torch)
    if cs is None or len(cs) < S + 1:                          # (the main line hands over ITS streams: every new stream takes another slot of the few hardware queues)
        cs = M.concurrent_streams(dev, S + 1)
    streams, nms_stream = cs[:S], cs[S]

    def loop(n):
        pending, dets = [], None
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]), torch.no_grad():
                pending.append(M.non_max_suppression_async(model(xs[k], slot=k)[0], 0.03, 0.65, multi_label=True, side=nms_stream))
            if len(pending) > S:
                dets = pending.pop(0).result()
        for h in pending:
            dets = h.result()
        return dets
    loop(48 + warmup)                                          # allocator pools + warm-up (untimed)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dets = loop(steps)
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    with torch.no_grad():
        for _ in range(5):
            model(xs[0])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            model(xs[0])
        torch.cuda.synchronize(dev)
    fwd = 1e3 * (time.perf_counter() - t0) / steps
    plan = model.plan_for(xs[0])
    tot = sum(plan.algorithmic_bytes(i) for i in range(len(plan.ops)))
    return {"metric": "images/sec MAF-YOLO-%s 640x640 bs=%d infer (Model.forward + non_max_suppression)" % (scale, batch), "value": round(batch * 1e3 / ms, 1), "unit": "images/s",
            "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup, "untimed_steps_before_the_timed_region": 48 + warmup, "batches_in_flight": S,
            "forward_only_ms": round(fwd, 4), "launches_per_forward": len(plan.ops), "whole_forward_hbm_frac": round(tot / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_GB_per_forward": round(tot / 1e9, 4), "detections_per_image_mean": round(float(np.mean([d.shape[0] for d in dets])), 1),
            "tiles": os.path.relpath(frozen[0], ROOT) if frozen else "timed at start-up"}


def nms_leg(torch, M, dev, pred, conf, iou, reps=40):
    """non_max_suppression ALONE on one batch of predictions (yolov6/utils/nms.py:31-105 at the settings of tools/eval.py: multi_label, max_det 300), milliseconds per
    batch between two events on one stream, for two class distributions of the SAME boxes and scores:
      synthetic_head      the timed workload's own prediction — its calibrated random head puts the ~2 000 candidates per image into two classes, so every image takes
                          the ALL-PAIRS path (the worst case of the design: one kept-list scan over all candidates), also timed in rounds 2-5's matrix form;
      classes_spread_80   the same tensor with every anchor's class scores rotated by (7 x anchor index) mod 80: the candidates spread over the 80 classes the way a
                          trained detector's do (evaler.py:178 sees a few dozen per class) and nms_sort_kernel sends the images down the PER-CLASS path
                          (nms_cscan_kernel: 80 short independent scans).
    `images_on_per_class_path` is read back from the workspace's per-image flags after the last call."""
    from maf_yolo_amd import nms as nms_mod, lib
    B, N, no = pred.shape
    nc = no - 5
    rot = (torch.arange(nc, device=dev)[None, :] + 7 * torch.arange(N, device=dev)[:, None]) % nc
    spread = pred.clone()
    spread[..., 5:] = torch.gather(pred[..., 5:], 2, rot[None].expand(B, N, nc))
    st = torch.cuda.current_stream(dev)

    def timed(p, matrix=False):
        keep, nms_mod.MATRIX_PATH = nms_mod.MATRIX_PATH, matrix
        try:
            for _ in range(5):
                rows, idx, cnt = nms_mod.nms_raw(p, conf, iou, multi_label=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                rows, idx, cnt = nms_mod.nms_raw(p, conf, iou, multi_label=True)
            e1.record(st)
            torch.cuda.synchronize(dev)
        finally:
            nms_mod.MATRIX_PATH = keep
        ws = nms_mod._workspace(dev, st, B, N, nc, 0)
        flags = ws[:B * lib.NMS_CNT_STRIDE * 4].view(torch.int32).view(B, lib.NMS_CNT_STRIDE)      # per image one 256-byte line: [0] candidates, [1] per-class path taken (csrc/nms.hip)
        return e0.elapsed_time(e1) / reps, int((flags[:, 1] != 0).sum()), float(flags[:, 0].float().mean()), float(cnt.float().mean())
    out = {}
    ms, pc, cand, det = timed(pred, matrix=False)
    ms_m = timed(pred, matrix=True)[0]
    out["synthetic_head"] = {"ms_per_batch": round(ms, 4), "images_on_per_class_path": pc, "candidates_per_image": round(cand, 1), "detections_per_image": round(det, 1),
                             "path": "all-pairs: kept-list scan (csrc/nms.hip nms_greedy_kernel: one workgroup per image — the form the timed loop's side-stream NMS takes)",
                             "matrix_form_ms_per_batch": round(ms_m, 4),
                             "matrix_form": "n x n suppression matrix on the whole chip + row scan: faster ALONE up to ~32 images, what a synchronous non_max_suppression() call takes (nms._matrix_form)"}
    ms, pc, cand, det = timed(spread)
    out["classes_spread_80"] = {"ms_per_batch": round(ms, 4), "images_on_per_class_path": pc, "candidates_per_image": round(cand, 1), "detections_per_image": round(det, 1),
                                "path": "per-class scans (nms_cscan_kernel)" if pc == B else "mixed"}
    out["note"] = "NMS alone, one stream, %d calls between two events; same boxes and scores in both rows, class axis rotated per anchor in the second" % reps
    return out


def host_cores():
    """Host cores this process may use: the scheduler affinity, capped by the cgroup CPU quota (the GPU box shows 256 CPUs, the container gets 16)
    and by 64 (oneDNN convs stop scaling beyond)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    return min(cores, 64)


def train_cpu_baseline(args, torch, M):
    """The same train step (train-form module tree in plain torch + the oracle's ComputeLoss restatement, fp32, autograd, SGD) on the host cores: a
    bounded sample (one warm-up + timed steps of 4 images until ~15 s have passed)."""
    from maf_yolo_amd import synth
    from oracle import maf_oracle as O
    threads = host_cores()
    torch.set_num_threads(threads)
    m = M.Model(args.scale)
    m.load_state_dict(synth.synth_state_dict(m, args.scale, 0))
    m = m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
    bs = 2
    x = synth.synth_images(bs, 640, seed=1)
    g = torch.Generator().manual_seed(100)
    wh = torch.rand(7 * bs, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(7 * bs, 2, generator=g) * (1 - wh)
    t = torch.cat([torch.arange(bs).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * bs, 1), generator=g).float(), ctr, wh], 1)

    def step():
        (feats, cls, reg), _ = m(x)
        loss = O.compute_loss([tuple(f.shape[-2:]) for f in feats], cls, reg, t, img_size=640)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    t_w = time.perf_counter()
    step()                                                      # warm-up (also tells how long a step takes)
    t_w = time.perf_counter() - t_w
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 + t_w < 20.0 and n < 20):     # bounded: at least one timed step, about 20 s of CPU work
        step(); n += 1
    el = time.perf_counter() - t0
    return {"value": round(bs * n / el, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d train steps of %d images (3x640x640 fp32): train-form module tree in plain torch + oracle.compute_loss, autograd, SGD; %.1f s wall, %d torch threads" % (n, bs, el, threads)}


# kernels (demangled, as tools/pmc_traffic.py writes them) behind a `roofline.by_kind` entry of the train step: one "launch" of the kind is one
# launch of EACH of them (a BatchNorm call = statistics kernel + apply kernel)
# (the profiler leaves these anonymous-namespace templates mangled: bn_stats_kernelIDF16_Lb1ELb0E = <_Float16, BWD = true, RES = false>)
_TRAIN_KIND_KERNELS = {"bn_act_backward": r"bn_(stats|apply)_kernel(<_Float16, true|IDF16_Lb1)", "bn_act_forward": r"bn_(stats|apply)_kernel(<_Float16, false|IDF16_Lb0)"}


def train_pmc_traffic(kind, scale):
    """HBM bytes per launch of `kind` from the PMC passes of a training run committed under profiles/ (tools/profile_round.sh: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs, counters only; tools/pmc_traffic.py: (FETCH_SIZE*2 + WRITE_SIZE)*1024) — n at batch 32 only."""
    import re
    path = [p_ for p_ in (os.path.join(ROOT, "profiles", f_) for f_ in ("round6_train_pmc_traffic.json", "round5_train_pmc_traffic.json", "round4_train_pmc_traffic.json", "round3_train_pmc_traffic.json", "round2_train_pmc_traffic.json")) if os.path.exists(p_)]
    path = path[0] if path else ""
    pat = _TRAIN_KIND_KERNELS.get(kind)
    if pat is None or scale != "n" or not os.path.exists(path):
        return None, None
    pmc = json.load(open(path))
    rows = [v for k, v in pmc.items() if re.search(pat, k)]
    calls = sum(v["launches"] for k, v in pmc.items() if re.search(pat.replace("(stats|apply)", "stats"), k))
    if not rows or not calls:
        return None, None
    total = sum(v["traffic_bytes"] * v["launches"] for v in rows)
    return int(total / calls), "profiles/%s: (FETCH_SIZE*2 + WRITE_SIZE)*1024 summed over the kind's kernels, per call (n, batch 32)" % os.path.basename(path)


def train_leg(args, torch, M, dev, rank, world, dist, scale, batch, steps, warmup, full):
    """The DDP train step of BASELINE configs[2]/[3] (yolov6/core/engine.py:141-167, 375-391, 477-489), timed with the bench protocol (barrier + device
    sync on both sides, MAX over ranks): forward (autocast fp16) + ComputeLoss + backward + the gradient exchange + fused SGD + EMA on a fixed
    synthetic batch per rank, 7 boxes per image.  EVERY world size — 1 included — runs the same schedule: maf_yolo_amd.GradExchange (weight
    gradients -> flat buckets on the weight-gradient stream, one RCCL all-reduce per bucket launched from that stream, the main stream waits
    once at the end of backward); at N = 1 the collectives are the only thing left out.  `--ddp` swaps in torch's DistributedDataParallel
    (which forces a per-layer join of the weight-gradient stream) for an A/B.  Returns the dict of results on rank 0 (None elsewhere)."""
    from maf_yolo_amd import synth, train_ops
    import torch.nn.functional as F
    if args.torch_convs:                       # A/B: same module tree, convs through F.conv2d (MIOpen)
        train_ops.conv1x1 = lambda x, w, bias=None: F.conv2d(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))
        train_ops.dwconv = lambda x, w: F.conv2d(x, w.to(x.dtype), None, 1, w.shape[-1] // 2, 1, x.shape[1])
    stats0 = dict(train_ops.stats)
    model = M.Model(scale)
    model.load_state_dict(synth.synth_state_dict(model, scale, 0))
    model = model.to(dev).train()
    if getattr(args, "no_tape", False):
        model.step_tape = False                # A/B: every step issued op by op from the autograd Functions (the path the step tape records)
    net, ex = model, None
    if args.ddp and world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], gradient_as_bucket_view=True)
    elif not args.ddp:
        # --rccl1: a ONE-rank RCCL group and force_collectives — the all-reduces (AVG, async, from the weight-gradient stream) are really
        # issued; what they cost on one GPU (a self-reduce per bucket) is reported as all_reduce.exposed_all_reduce_ms
        ex = M.GradExchange(model, force_collectives=bool(getattr(args, "rccl1", False)) and dist is not None)
    # build.py:12-33 grouping (BatchNorm weights and biases without decay), lr0 scaled like engine.py: fused SGD keeps the AMP inf check on the device
    opt = M.build_optimizer(model, lr0=0.01 / 64 * batch * world, momentum=0.937, weight_decay=5e-4, fused=not args.no_fused_sgd)
    scaler = M.GradScaler("cuda")                 # torch.amp.GradScaler with the inf check as one native launch (solver.GradScaler; MAF_INF_CHECK_NATIVE=0: the framework's)
    ema = M.ModelEMA(model) if rank == 0 and not args.no_ema else None          # engine.py:67: the main process keeps the weight average
    B = batch
    x = synth.synth_images(B, 640, seed=1 + rank).to(dev)          # engine.py:426: float images / 255
    g = torch.Generator().manual_seed(100 + rank)
    nbox = 7 * B
    wh = torch.rand(nbox, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(nbox, 2, generator=g) * (1 - wh)
    targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (nbox, 1), generator=g).float(), ctr, wh], 1).to(dev)
    crit = M.ComputeLoss(ori_img_size=640, warmup_epoch=0)   # steady state of the schedule (epoch >= 3): the task-aligned assigner

    def step():
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = net(x)
        if args.surrogate_loss:
            loss = (cls.float().mean() + reg.float().pow(2).mean()) * world
        else:
            loss = crit((feats, cls, reg), targets, 0, 0)[0] * world            # engine.py:161-162 (loss scaled by the world size)
        if ex is not None:
            ex.zero_grad()                                                      # one memset per bucket; p.grad stays a view of its bucket
        else:
            opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()                                           # ex.finish() runs as the autograd engine's final callback
        scaler.step(opt)
        scaler.update()
        if ema is not None:
            ema.update(net)                                                     # engine.py:389-390
        return loss

    def timed(fn, n):
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)
        t0_ = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0_
        if dist is not None:
            dist.barrier()
            t_ = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el = t_.item()
        return el

    for _ in range(warmup):
        loss = step()
    last = {}

    def one():
        last["loss"] = step()
    elapsed = timed(one, steps)
    loss = last["loss"]
    launches = {k: v - stats0.get(k, 0) for k, v in train_ops.stats.items()}    # before the CPU baseline runs the same module tree on the host
    # ---- exposed (non-overlapped) all-reduce time (BASELINE configs[3], yolov6/core/engine.py:477-489): the same steps with the collectives
    # switched off (no_sync) — the difference is what the exchange adds to a step after its overlap with the remaining backward
    comm = None
    if world > 1 or (ex is not None and ex.force):
        def nosync():
            with (ex.no_sync() if ex is not None else net.no_sync()):
                step()
        k2 = max(5, steps // 2)
        el_ns = timed(nosync, k2)
        grad_bytes = sum(p_.numel() * 4 for p_ in model.parameters() if p_.requires_grad)
        comm = {"ms_per_step_without_all_reduce": round(1e3 * el_ns / k2, 3), "exposed_all_reduce_ms": round(1e3 * (elapsed / steps - el_ns / k2), 3),
                "gradient_bytes_per_step": grad_bytes, "steps": k2,
                "buckets": None if ex is None else [int(b_.flat.numel() * 4) for b_ in ex.buckets],
                "note": ("same steps under GradExchange.no_sync(); one RCCL all-reduce (AVG) per flat fp32 bucket, launched from the weight-gradient stream as soon as "
                         "the bucket's last gradient has been issued" if ex is not None else
                         "same steps under DistributedDataParallel.no_sync(); bucket all-reduces from the autograd hooks")}
    # ---- roofline of the dominant training kernel: one more step with HIP events around every native launch (train_ops.profile)
    roof = None
    if not args.torch_convs:
        # EVERY rank runs this step (it carries the gradient all-reduce: a step on rank 0 alone would wait for its peers forever);
        # only rank 0 records the events
        model.step_tape = False                # the event pairs need the eager path (one autograd Function per launch); one untimed eager step first: after the
        step()                                 # replayed steps its weight staging plan and Python paths are cold, and a host-bound step inflates what the event pairs see
        if rank == 0:
            train_ops.profile = {}
        step()
        torch.cuda.synchronize(dev)
    if rank == 0 and not args.torch_convs:
        prof = train_ops.profile_collect()
        train_ops.profile = None
        tot = sum(v[0] for v in prof.values())
        kinds = sorted(prof.items(), key=lambda kv: -kv[1][0])
        k0, (ms0, by0, n0) = kinds[0]
        traffic, traffic_src = train_pmc_traffic(k0, scale)
        roof = {"bound": "hbm", "kernel": k0, "launches_per_step": n0, "avg_launch_ms": round(ms0 / n0, 5), "bytes_per_launch": int(by0 / n0),
                "achieved": round(by0 / ms0 / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by0 / ms0 / 1e6 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "share_of_native_kernel_time": round(ms0 / tot, 4), "native_kernel_ms_per_step": round(tot, 3)}
        # the whole step against the HBM roofline: the algorithmic bytes of every native launch of the step (each kernel reads its inputs once and writes its outputs
        # once: the per-launch figures above, summed over both streams) over the step's wall time
        step_gb = sum(v[1] for v in prof.values()) / 1e9
        roof["whole_step"] = {"algorithmic_GB": round(step_gb, 3), "ms_per_step": round(1e3 * elapsed / steps, 3),
                              "hbm_frac": round(step_gb / (elapsed / steps) / HBM_PEAK_GBS, 4), "native_launches": int(sum(v[2] for v in prof.values()))}
        if full:
            roof["by_kind"] = {k: {"ms": round(v[0], 3), "launches": v[2], "achieved_GBs": round(v[1] / v[0] / 1e6, 1)} for k, v in kinds}
    final = float(loss.detach())
    tapes = [e_[1] for e_ in getattr(model, "_tapes", {}).values() if e_[1] is not None]
    tape_info = {"replayed_steps": int(launches.get("tape_replays", 0)), "ready": bool(tapes and tapes[0].ready), "dropped_because": tapes[0].failed if tapes else None,
                 "launches_forward": tapes[0].n.get("fwd") if tapes else None, "launches_backward": tapes[0].n.get("bwd") if tapes else None,
                 "what": "maf_yolo_amd/tape.py: the step's C-ABI calls recorded once per batch shape (third step) and replayed by maf_tape_run, eager launches from one C loop"}
    exs, nb = None, 0
    if ex is not None:
        exs, nb = dict(ex.stats), len(ex.buckets)
        ex.close()
    if not math.isfinite(final):
        raise SystemExit("bench.py train leg: the loss is not finite after %d steps — the run is invalid" % (warmup + steps))
    if rank != 0:
        return None
    cpu = None
    if full and not args.no_cpu_baseline:
        cpu = train_cpu_baseline(args, torch, M)
    fallback = int(launches.get("fallback", 0) + launches.get("torch_bn", 0) + launches.get("torch_maxpool", 0) + launches.get("framework_wgrad_fp32", 0))
    return {"metric": "train images/sec MAF-YOLO-%s 640x640 bs=%d/GPU DDP (fwd+bwd+all-reduce+SGD+EMA, AMP fp16)" % (scale, B),
            "value": round(world * B * steps / elapsed, 1), "images_per_s": round(world * B * steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "MAF-YOLO-%s train-form, %d x 3x640x640 per GPU, %s" % (scale, B, "surrogate loss over all head outputs" if args.surrogate_loss else "ComputeLoss (HIP task-aligned assigner + VFL/GIoU/DFL), 7 boxes/image"),
                       "global_batch": B * world, "parallelism": "ddp%d" % world,
                       "gradient_exchange": ("torch DistributedDataParallel" if ex is None and world > 1 else "none (plain autograd)" if ex is None else
                                             "maf_yolo_amd.GradExchange: %d flat fp32 buckets filled on the weight-gradient stream, all-reduce per bucket from that stream%s"
                                             % (nb, "" if world > 1 else " (one-rank RCCL group: the collectives ARE issued, --rccl1)" if ex.force else " (world size 1: same schedule, no collective)")),
                       "exchange_stats": exs, "step_tape": tape_info,
                       "convs": "torch/MIOpen" if args.torch_convs else "HIP kernels for every conv (1x1, depth-wise, 3x3 s2, 1x1 s2: forward, data gradient, weight gradient) and BatchNorm(train)+activation",
                       "native_launches": launches, "fallback": fallback, "final_loss": round(final, 5)},
            "native_launches": int(sum(v for k, v in launches.items() if k.startswith("native_"))), "fallback": fallback, "final_loss": round(final, 5),
            "ranks_seen": 1 if dist is None else int(dist.get_world_size()),
            "roofline": roof, "cpu_baseline": cpu, "all_reduce": comm}


def train_mode(args, torch, M, dev, rank, world, dist):
    res = train_leg(args, torch, M, dev, rank, world, dist, args.scale, args.batch, args.steps, args.warmup, True)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): replace this process by the launch the docstring names —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` — one rank
    per GPU (tools/train.py:109-114 reads the same LOCAL_RANK / RANK / WORLD_SIZE).  Runs before torch is imported; does not return."""
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def rendezvous_only(args, rank, world):
    """--rendezvous-only: the launch, the process group and the timing protocol of the N > 1 run (barrier on both sides, MAX over ranks, ONE line from rank 0)
    with nothing between the barriers — no device, no kernel.  tests/test_host_logic.py runs `bench.py --gpus 2 --dist-backend gloo --rendezvous-only`
    on the CPU-only build container; it is not a measurement and says so in the line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(args.dist_backend if args.dist_backend != "nccl" or torch.cuda.is_available() else "gloo", init_method="env://")
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))                                # rank r "works" 10 (r + 1) ms: the MAX over ranks must be the last rank's
    el = time.perf_counter() - t0
    dist.barrier()
    t = torch.tensor([el], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "rendezvous only (no kernel ran): launch + process group + timing protocol of bench.py --gpus N", "value": None, "n_gpus": world,
                          "ranks_seen": int(seen.item()), "world_size": dist.get_world_size(), "max_over_ranks_s": round(t.item(), 4), "rank0_s": round(el, 4),
                          "backend": dist.get_backend(), "launched_by": "self_launch" if os.environ.get("TORCHELASTIC_RUN_ID") is not None else "env"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (BASELINE.md §4: >= 100; 300 = a 0.4 s region: one 20 ms stall of the box no longer moves the value by 15 %%)")
    ap.add_argument("--warmup", type=int, default=30, help="untimed warm-up steps (BASELINE.md §4: >= 20)")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (BASELINE configs[1]: 32)")
    ap.add_argument("--scale", default="n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--nms-filter", action="store_true", help="A/B: the forward's head tails collect the NMS candidates (Model.nms_filter) instead of non_max_suppression's own pass over the prediction")
    ap.add_argument("--nms-cus", type=int, default=0, help="A/B: run the NMS stream on this many compute units only (hipExtStreamCreateWithCUMask); 0 = the whole chip")
    ap.add_argument("--per-op", action="store_true", help="also print the per-op table to stderr")
    ap.add_argument("--lanes", type=int, default=-1, help="engine streams: 0 one stream, 1 heads on side streams, 2 heads + neck side convs (default: the model's setting)")
    ap.add_argument("--fuse", type=int, default=-1, help="1/0: force the fused DepthBottleneckUni kernel on/off (default: the model's setting)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight in the timed serving loop: step i runs on HIP stream i %% S with its own activation arena "
                         "(1 = one stream; the NMS of a batch still overlaps the next forward)")
    ap.add_argument("--tune-file", default=None,
                    help="JSON of autotuned tiles: loaded if it exists (no re-timing: profiler passes run the same kernels as the bench), written after tuning; "
                         "default: profiles/round2_tune.json if present; 'none' = time every layer afresh")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[2]/[3] instead: DDP training step (train-form graph, AMP fp16, SGD), images/s; use with --scale s|m --batch 32|16")
    ap.add_argument("--surrogate-loss", action="store_true", help="with --train: mean over the head outputs instead of ComputeLoss")
    ap.add_argument("--torch-convs", action="store_true", help="with --train: run the 1x1 / depth-wise convs on stock PyTorch-ROCm (MIOpen) for an A/B")
    ap.add_argument("--ddp", action="store_true", help="--train A/B: torch's DistributedDataParallel instead of maf_yolo_amd.GradExchange (N > 1; at N = 1: plain autograd)")
    ap.add_argument("--no-train-leg", action="store_true", help="leave the short training leg (`train` object: n, bs 32/GPU) out of the default line")
    ap.add_argument("--no-extra-legs", action="store_true", help="leave the compact legs of the other BASELINE configs (train_s, train_m, latency_m, infer_s, infer_m) out of the default line")
    ap.add_argument("--train-steps", type=int, default=30, help="timed steps of the training leg of the default line (after 6 warm-up steps: the first ones time the conv variants per shape)")
    ap.add_argument("--no-tape", action="store_true", help="--train A/B: no step tape (maf_yolo_amd/tape.py): every step issued op by op from Python")
    ap.add_argument("--no-ema", action="store_true", help="--train A/B: leave the ModelEMA update of rank 0 out of the step")
    ap.add_argument("--no-fused-sgd", action="store_true", help="--train A/B: torch.optim.SGD's default (foreach) implementation; GradScaler.step then syncs the host every step")
    ap.add_argument("--rccl1", action="store_true", help="--train at N = 1: initialise a one-rank RCCL group and issue the bucket all-reduces anyway (GradExchange(force_collectives=True))")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the N > 1 code on one GPU)")
    ap.add_argument("--rendezvous-only", action="store_true", help="N > 1: launch, process group, barrier / MAX-over-ranks protocol and the JSON line only — no device work (the CPU test of the launch path)")
    ap.add_argument("--latency", action="store_true",
                    help="BASELINE configs[4] instead: bs=1 forward replayed from a hipGraph + fused NMS, p50/p99 latency (use with --scale m)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import maf_yolo_amd as M
    from maf_yolo_amd import lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                 # `python bench.py --gpus N` as typed for N = 1: becomes the N-rank launch (does not return)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with --nproc-per-node %d, or with no launcher at all" % (args.gpus, world, args.gpus))
    if args.rendezvous_only:
        return rendezvous_only(args, rank, world)
    # MAF_BENCH_ONE_DEVICE=1 + --dist-backend gloo: every rank on cuda:0 with the exchange over gloo — NOT a measurement, a way to run the N > 1 code
    # paths (barriers, MAX over ranks, DDP hooks, the no_sync re-timing) on a 1-GPU box, where RCCL refuses two ranks on one device
    one_dev = os.environ.get("MAF_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_dev else local_rank)
    dev = torch.device("cuda", 0 if one_dev else local_rank)
    dist = None
    if world == 1 and args.rccl1:
        import socket
        import torch.distributed as dist
        so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    if world > 1:
        import torch.distributed as dist
        import datetime
        # a rank that dies alone must not leave its peers inside a collective for the default 10 minutes: 3 minutes, then the job fails loudly
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=dev, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(args.dist_backend, init_method="env://", timeout=datetime.timedelta(seconds=180))

    if args.latency:
        return latency_mode(args, torch, M, dev)
    if args.train:
        return train_mode(args, torch, M, dev, rank, world, dist)

    # ---- model + synthetic inputs
    from maf_yolo_amd import synth
    model = M.Model(args.scale)
    sd = synth.synth_state_dict(model, args.scale, 0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.autotune = not args.no_autotune      # per-layer MFMA tile selection when the plan is built (outside the timed region)
    if args.fuse >= 0:
        model.fuse_bottlenecks = {0: False, 1: True}.get(args.fuse, args.fuse)
    if args.lanes >= 0:
        model.multi_stream = args.lanes
    from maf_yolo_amd import engine as _engine
    # tile choices (which (pixels x channels) cut, which kernel variant per layer) measured on an MI355X and frozen in the repo are the
    # default starting point: layer signatures that are not in the file are still timed here.  --tune-file none = time everything afresh.
    frozen = [p_ for p_ in (os.path.join(ROOT, "profiles", f_) for f_ in ("round6_tune.json", "round5_tune.json", "round4_tune.json", "round3_tune.json")) if os.path.exists(p_)]
    if args.tune_file is None and frozen:
        args.tune_file = frozen[0]
    if args.tune_file == "none":
        args.tune_file = None
    if args.tune_file and os.path.exists(args.tune_file):
        _engine.load_tune_cache(args.tune_file)
    B = args.batch
    x = synth.synth_images(B, 640, seed=1 + rank).to(dev).half()
    shift = calibrate_cls_bias(model, x, 2000, M, torch)
    conf, iou = 0.03, 0.65
    with torch.no_grad():
        cand = (model(x)[0][..., 5:] > conf).sum((1, 2))
    if args.nms_filter:
        # A/B: the head tails append the NMS candidates while they hold the class scores (Model.nms_filter -> maf_engine_run_filtered;
        # non_max_suppression then skips its own pass over the 91 MB prediction).  Same detections bit for bit
        # (tests/test_gpu_model.py:test_candidate_filter_inside_the_forward_gives_the_same_detections); measured: no gain in this loop (the
        # candidate pass runs on the NMS stream under the next forward anyway, the forward grows by ~10 us) — off by default.
        model.nms_filter = conf
    cand_mean, cand_max = float(cand.float().mean()), int(cand.max())
    if args.tune_file and rank == 0 and not os.path.exists(args.tune_file):
        _engine.save_tune_cache(args.tune_file)

    def step():
        with torch.no_grad():
            pred = model(x)[0]
        return M.non_max_suppression(pred, conf, iou, multi_label=True)

    def fwd_only():
        with torch.no_grad():
            return model(x)[0]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    S = max(1, args.inflight)
    xs = [x] + [synth.synth_images(B, 640, seed=101 + 7 * k + rank).to(dev).half() for k in range(1, S)]
    # streams that share a hardware queue run their kernels one after the other: take streams that demonstrably overlap (streams.py)
    cs = M.concurrent_streams(dev, S + 1)                      # S forward streams + the NMS stream, on different hardware queues
    streams, nms_stream = (cs[:S] if S > 1 else [torch.cuda.current_stream(dev)]), cs[S]
    if args.nms_cus > 0:
        # the NMS stream confined to a slice of the chip (streams.masked_stream): its kernels no longer spread over every CU beside the next forward's first kernels
        from maf_yolo_amd.streams import masked_stream
        nms_stream = masked_stream(dev, args.nms_cus)
    probe0 = M.concurrent_streams.last_ratio

    def pipelined(n):
        """n steps of the serving loop: every step = one forward + one NMS of a batch.  Up to S batches are in flight: step i runs on
        stream i % S with that slot's own activation arena (Model.forward(slot=)), its NMS goes to a side stream, and its result is
        collected when the slot comes round again.  Same work per step as step(): nothing is skipped, all n forwards and all n NMS
        results are complete when this returns."""
        pending, dets = [], None
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]), torch.no_grad():
                pred_i = model(xs[k], slot=k)[0]                  # every slot has its own resident input batch (not one MALL-warm tensor)
                pending.append(M.non_max_suppression_async(pred_i, conf, iou, multi_label=True, side=nms_stream))
            if len(pending) > S:                                   # collect batch i - S once batch i is queued
                dets = pending.pop(0).result()
        for h in pending:
            dets = h.result()
        return dets

    # ---- the timed region: K steps of forward + NMS, software-pipelined across steps (the host-side hand-over of the NMS result — a
    #      count read-back and 32 slices — otherwise idles the GPU for 0.1-0.3 ms per step, and makes the number follow host jitter)
    # set-up, not part of the W warm-up steps: the first few dozen iterations of the loop grow the caching allocator's pools (prediction
    # tensors handed to the NMS stream return to the forward stream's pool late), and every hipMalloc that causes stalls the device —
    # measured as an occasional 2.5 ms/step first run of an otherwise 1.67 ms/step loop.  Run the loop until the pools are settled.
    POOL_SETTLE_STEPS = 48                        # reported in the line as `pool_settle_steps`: untimed, on top of the W warm-up steps
    pipelined(POOL_SETTLE_STEPS)
    if args.warmup:
        dets = pipelined(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    dets = pipelined(args.steps)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    from maf_yolo_amd.streams import overlap_ratio
    probe1 = overlap_ratio(cs)
    arenas = [model.plan_for(x, slot=k).arena.data_ptr() for k in range(S)]

    # ---- one batch in flight (one stream; only the NMS of batch i overlaps the forward of batch i+1), rank-local
    one_ms = None
    if S > 1:
        S_keep, streams_keep = S, streams
        S, streams = 1, [streams_keep[0]]                   # a real stream, not the legacy default one (which synchronises with every other stream)
        sync_all()
        t0 = time.perf_counter()
        dets = pipelined(args.steps)
        torch.cuda.synchronize(dev)
        one_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        S, streams = S_keep, streams_keep

    # ---- the same steps strictly one after the other (forward, NMS, result on the host, next forward), rank-local
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dets = step()
    torch.cuda.synchronize(dev)
    seq_ms = 1e3 * (time.perf_counter() - t0) / args.steps

    # ---- forward-only rate (same protocol, rank-local) and per-kernel roofline (rank 0)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd_only()
    torch.cuda.synchronize(dev)
    fwd_ms = 1e3 * (time.perf_counter() - t0) / args.steps

    nms_alone = None
    if rank == 0:
        with torch.no_grad():
            nms_alone = nms_leg(torch, M, dev, model(x)[0], conf, iou)
    out = None
    if rank == 0:
        plan = model.plan_for(x)
        pred = torch.empty(B, plan.A, 5 + plan.nc, dtype=torch.float32, device=dev)
        reps = 10
        acc = np.zeros(len(plan.ops))
        plan.run_timed(x, pred)
        for _ in range(reps):
            acc += np.array(plan.run_timed(x, pred))
        per_op_ms = acc / reps
        groups = {}
        for i, ms in enumerate(per_op_ms):
            g = groups.setdefault(plan.kernel_name(i), dict(ms=0.0, n=0, bytes=0, flops=0))
            g["ms"] += ms; g["n"] += 1; g["bytes"] += plan.algorithmic_bytes(i); g["flops"] += plan.flops(i)
        # The dominant kernel = the kernel TEMPLATE with the largest share of forward time (its instantiations are tile / variant
        # choices of one piece of code, picked per layer by the autotuner); every instantiation is listed beside it.
        fams = {}
        for k, g in groups.items():
            f = fams.setdefault(k.split("<")[0], dict(ms=0.0, n=0, bytes=0, flops=0, inst=[]))
            f["ms"] += g["ms"]; f["n"] += g["n"]; f["bytes"] += g["bytes"]; f["flops"] += g["flops"]; f["inst"].append(k)
        name, gd = max(fams.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = gd["ms"] / gd["n"]
        bytes_per_launch = gd["bytes"] / gd["n"]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        tot_bytes = sum(plan.algorithmic_bytes(i) for i in range(len(plan.ops)))
        tot_flops = sum(plan.flops(i) for i in range(len(plan.ops)))
        fwd_img_s = B / (fwd_ms * 1e-3)
        traffic, traffic_src, pmc = None, None, {}
        try:                                   # PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs) committed under profiles/
            pmc_file = [f_ for f_ in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f_))][0]
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            have = [k for k in gd["inst"] if k in pmc]
            if have:                           # per launch of the template: instantiations weighted by their launches in this forward
                traffic = int(sum(pmc[k]["traffic_bytes"] * groups[k]["n"] for k in have) / sum(groups[k]["n"] for k in have))
                traffic_src = "profiles/%s: (FETCH_SIZE*2 + WRITE_SIZE)*1024 per launch, gfx950 FETCH_SIZE x2 correction" % pmc_file
        except Exception:
            pass
        # the same family's average launch duration in the committed rocprofv3 --kernel-trace --stats summary (tools/profile_round.sh runs this very
        # command under the profiler): beside the live event figure, which carries ~2.5 us of event-pair overhead per launch
        rocprof_avg_ms, rocprof_src = None, None
        try:
            import csv as _csv
            stats_file = [f_ for f_ in ("round6_kernel_stats.csv", "round5_kernel_stats.csv", "round4_kernel_stats.csv", "round3_kernel_stats.csv") if os.path.exists(os.path.join(ROOT, "profiles", f_))][0]
            tot_ns = calls = 0
            for row in _csv.DictReader(open(os.path.join(ROOT, "profiles", stats_file))):
                if name in row["Name"]:
                    tot_ns += int(row["TotalDurationNs"]); calls += int(row["Calls"])
            if calls:
                rocprof_avg_ms, rocprof_src = round(tot_ns / calls / 1e6, 5), "profiles/%s (%d calls)" % (stats_file, calls)
        except Exception:
            pass
        insts = []
        for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            a_gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
            insts.append(dict(kernel=k, launches=g["n"], avg_launch_ms=round(g["ms"] / g["n"], 5), share_of_forward=round(g["ms"] / per_op_ms.sum(), 4),
                              achieved_GBs=round(a_gbs, 1), hbm_frac=round(a_gbs / HBM_PEAK_GBS, 4),
                              mfma_frac=round(g["flops"] / (g["ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                              traffic=pmc.get(k, {}).get("traffic_bytes")))
        k_big, g_big = max(groups.items(), key=lambda kv: kv[1]["ms"])
        big_gbs = g_big["bytes"] / (g_big["ms"] * 1e-3) / 1e9
        largest = dict(kernel=k_big, launches=g_big["n"], avg_launch_ms=round(g_big["ms"] / g_big["n"], 5), share_of_forward=round(g_big["ms"] / per_op_ms.sum(), 4),
                       bytes_per_launch=int(g_big["bytes"] / g_big["n"]), achieved=round(big_gbs, 1), frac=round(big_gbs / HBM_PEAK_GBS, 4), traffic=pmc.get(k_big, {}).get("traffic_bytes"))
        # SURVEY.md 8(d) layer-granular byte model (every conv block of the UNFUSED deploy graph reads its input and writes its output once):
        # activations per image + weights once per batch; the plan's own model (`algorithmic_GB`) counts the fused graph, where the tensors
        # between fused layers never exist
        act_w = {"n": (174.5, 7.5), "s": (339.3, 17.1), "m": (679.3, 47.4)}[args.scale]
        layer_gb = (B * act_w[0] + act_w[1]) / 1e3
        roofline = dict(bound="hbm", kernel=name + "<...> (%d instantiations)" % len(gd["inst"]), launches_per_forward=gd["n"], avg_launch_ms=round(avg_ms, 5),
                        bytes_per_launch=int(bytes_per_launch), achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                        rocprof_avg_launch_ms=rocprof_avg_ms, rocprof_source=rocprof_src,
                        frac_rocprof=None if not rocprof_avg_ms else round(bytes_per_launch / (rocprof_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        share_of_forward=round(gd["ms"] / per_op_ms.sum(), 4), largest_symbol=largest, top_instantiations=insts,
                        whole_forward=dict(algorithmic_GB=round(tot_bytes / 1e9, 4), layer_granular_GB=round(layer_gb, 4),
                                           hbm_frac_layer_granular=round(layer_gb / (fwd_ms * 1e-3) / HBM_PEAK_GBS, 4), GFLOP=round(tot_flops / 1e9, 2),
                                           sum_kernel_ms=round(float(per_op_ms.sum()), 4),
                                           hbm_frac=round(tot_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           hbm_frac_timed_region=round(tot_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           mfma_frac=round(tot_flops / (fwd_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 5)))
        if args.per_op:
            order = np.argsort(-per_op_ms)
            for i in order:
                gb = plan.algorithmic_bytes(i) / (per_op_ms[i] * 1e-3) / 1e9
                o = plan.ops[i]
                print("%-34s %-44s %8.4f ms %8.1f GB/s  %dx%d %d->%d k%d" % (plan.op_names[i], plan.kernel_name(i), per_op_ms[i], gb, o.H, o.W, o.Cin, o.Cout, o.ksize), file=sys.stderr)
            for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
                print("GROUP %-46s n=%2d total %8.4f ms  avg %8.4f ms  %8.1f GB/s" %
                      (k, g["n"], g["ms"], g["ms"] / g["n"], g["bytes"] / (g["ms"] * 1e-3) / 1e9), file=sys.stderr)

        cpu = None

        def cpu_leg():                                  # runs AFTER the training leg: 20 s of all host cores right in front of a host-bound train step cost it up to 15 %
            from oracle import maf_oracle as O          # the CPU restatement: checker / baseline only
            cores = host_cores()
            torch.set_num_threads(cores)
            dw = O.reparam(sd, args.scale)
            for m_ in model.backbone:          # same calibrated head as the GPU run
                if hasattr(m_, "cls_pred"):
                    key = "backbone.%d.cls_pred" % m_.i
                    dw[key] = (dw[key][0], m_.cls_pred.bias.detach().float().cpu())
            # batches of 8 and of 32, 3 warm-up + 5 timed runs each; `value` = the bs-32 rate (the metric's batch)
            runs = {}
            with torch.no_grad():
                for bs_ in (8, 32):
                    xb = torch.cat([t_.float().cpu() for t_ in xs])[:bs_] if bs_ > B else x[:bs_].float().cpu()
                    ts = []
                    for it in range(3 + 5):
                        t0 = time.perf_counter()
                        p = O.predict(dw, args.scale, xb)
                        O.non_max_suppression(p.numpy(), conf, iou, multi_label=True)
                        if it >= 3:
                            ts.append(time.perf_counter() - t0)
                    runs[bs_] = dict(images_per_s=round(xb.shape[0] / float(np.median(ts)), 2), best=round(xb.shape[0] / min(ts), 2), batch=int(xb.shape[0]), timed_runs=len(ts), warmup_runs=3,
                                     seconds=round(float(sum(ts)), 1))
            cpu = dict(value=runs[32]["images_per_s"], unit="images/s", cores=cores, kind="port",
                       sample="oracle.predict + oracle.non_max_suppression (fp32, %d torch threads) on 3x640x640 batches of 8 and 32: 3 warm-up + 5 timed runs each, median; "
                              "value = batch %d (%.1f s timed), batch 8: %.2f images/s" % (cores, runs[32]["batch"], runs[32]["seconds"], runs[8]["images_per_s"]),
                       runs={str(k): v for k, v in runs.items()})
            return cpu

        out = {"metric": "images/sec MAF-YOLO-%s 640x640 bs=%d infer (Model.forward + non_max_suppression)" % (args.scale, B),
               "value": round(value, 1), "unit": "images/s", "n_gpus": world, "ranks_seen": 1 if dist is None else int(dist.get_world_size()), "steps": args.steps, "warmup": args.warmup,
               "pool_settle_steps": POOL_SETTLE_STEPS, "untimed_steps_before_the_timed_region": POOL_SETTLE_STEPS + args.warmup,
               "timed_region_s": round(elapsed, 4),
               "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16", "data": "synthetic",
               "config": {"workload": "MAF-YOLO-%s deploy-form inference, %d x 3x640x640 fp16 per GPU resident in HBM, "
                                      "forward + NMS(conf 0.03, iou 0.65, multi_label); synthetic seeded weights, cls bias "
                                      "calibrated (%+.2f) to ~2000 candidates/img" % (args.scale, B, shift),
                          "batch_per_gpu": B, "global_batch": B * world, "parallelism": "replicas x%d (no collective)" % world,
                          "execution": "K steps of forward + NMS of one batch each; %d batches in flight (step i on HIP stream i %% %d with its own "
                                       "activation arena, its NMS on a side stream); all K forwards and K NMS results complete inside the timed region" % (S, S),
                          "batches_in_flight": S, "arena_base": ["0x%x" % a_ for a_ in arenas], "stream_overlap_probe": {"at_start": round(probe0, 2), "after_timed_region": round(probe1, 2),
                                                                           "meaning": "spin kernels on all serving streams at once / one alone: ~1 = distinct hardware queues"},
                          "tiles": ("per-layer tile / variant choices loaded from %s, the rest timed at start-up" % os.path.relpath(args.tune_file, ROOT)) if args.tune_file and os.path.exists(args.tune_file) else "every layer's tile / variant timed at start-up",
                          "nms_candidates_per_image": {"mean": round(cand_mean, 1), "max": cand_max},
                          "detections_per_image_mean": round(float(np.mean([d.shape[0] for d in dets])), 1)},
               "forward_only": {"ms_per_step": round(fwd_ms, 4), "images_per_s_per_gpu": round(B / (fwd_ms * 1e-3), 1)},
               "one_in_flight": None if one_ms is None else {"ms_per_step": round(one_ms, 4), "images_per_s_per_gpu": round(B / (one_ms * 1e-3), 1),
                                                              "note": "one stream, one arena: only the NMS of batch i overlaps the forward of batch i+1 (rank-local)"},
               "sequential": {"ms_per_step": round(seq_ms, 4), "images_per_s_per_gpu": round(B / (seq_ms * 1e-3), 1),
                              "note": "the same steps with no overlap: forward, NMS, result handed to the host, next forward (rank-local); the GPU idles "
                                      "during the host hand-over, so this one follows host jitter"},
               "nms": nms_alone,
               "parity_bar_of_the_timed_plan": "fp16 fused plan: every fused kernel within 2e-3 * max|ref| + 2e-3 of the unfused restatement (tests/test_gpu_fused_parity.py), end to end "
                                               ">= 294 of 300 reference detections per image matched at IoU >= 0.95 and |d score| <= 2e-3 on 2 x 640^2 (n; s >= 290, m >= 280), NMS rows and "
                                               "indices bit-identical to the reference fixtures; north_star's 1e-3 on boxes / scores is met by the fp32 plan (unfused kernels), not by this one",
               "roofline": roofline, "cpu_baseline": cpu}
    # ---- the training half of BASELINE's metric ("train imgs/s @1/2/4/8"): a short leg of the DDP train step (n, 32 images per GPU) inside the
    # same driver-timed run; `--train` is the long form (any scale / batch, per-kind roofline, CPU baseline)
    train = None
    if not args.no_train_leg:
        del dets
        model._plans = {}
        torch.cuda.empty_cache()
        train = train_leg(args, torch, M, dev, rank, world, dist, "n", 32, args.train_steps, 10, False)       # 30 timed steps after 10 warm-up steps: a 0.65 s region (12 after 6 moved by 2 ms per step with one host stall: 23.8 in the line, 21.8 alone, same box)
    # ---- every other BASELINE config at N = 1 in the same driver-timed run, compact: configs[2] / [3] at one GPU (s bs 32, m bs 16 per GPU),
    # configs[4] (m, bs 1, latency: eager launches and hipGraph replay), and the headline workload for s and m.  ~10 s each; --no-extra-legs leaves them out.
    extra = {}
    if not args.no_extra_legs and not args.no_train_leg:
        def leg(name, fn):
            t_ = time.perf_counter()
            try:
                r_ = fn()
            except Exception as e_:                                  # a leg must not take the headline line with it
                r_ = {"error": "%s: %s" % (type(e_).__name__, e_)}
            if isinstance(r_, dict):
                r_.pop("cpu_baseline", None)
                r_["leg_wall_s"] = round(time.perf_counter() - t_, 1)
            extra[name] = r_
            torch.cuda.empty_cache()
        # configs[2] (s, 32 images per GPU: global 64 at N = 2) and configs[3] (m, 16 per GPU: global 128 at N = 8) run at EVERY world size the driver asks for:
        # every rank takes part (the step carries the bucket all-reduces), rank 0 keeps the result with `all_reduce.exposed_all_reduce_ms`, the bucket sizes
        # and `ranks_seen`.  (A failing leg raises on every rank alike — same code, same shapes — so the try / except cannot leave a peer inside a collective
        # unless a rank fails alone; then the job dies on the collective's timeout instead of hanging.)
        leg("train_s", lambda: train_leg(args, torch, M, dev, rank, world, dist, "s", 32, 15, 8, False))
        leg("train_m", lambda: train_leg(args, torch, M, dev, rank, world, dist, "m", 16, 15, 8, False))
        if world == 1:                                             # rank-local legs: replicas would only repeat them
            leg("latency_m", lambda: latency_leg(torch, M, dev, "m", 300, 30))
            leg("infer_s", lambda: infer_leg(torch, M, dev, "s", 32, 100, 20, cs))
            leg("infer_m", lambda: infer_leg(torch, M, dev, "m", 32, 60, 20, cs))
    if rank == 0:
        if train is not None:
            train.pop("cpu_baseline", None)
        out["train"] = train
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_leg()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
