#!/usr/bin/env python3
"""bench.py — images/s of the MAF-YOLO-n hot path on MI355X (BASELINE.json metric, configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
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
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The serving loop keeps three batches in flight on three streams plus one for the NMS.  ROCm multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run one after the other: give them room.  (Must be set
# before the HIP runtime starts, i.e. before torch is imported; an explicit setting in the environment wins.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def latency_mode(args, torch, M, dev):
    """bs=1, 640x640 fp16: forward replayed from a captured hipGraph (fixed input/output buffers) + NMS, per-call latency."""
    import numpy as np
    from maf_yolo_amd import synth
    model = M.Model(args.scale)
    model.load_state_dict(synth.synth_state_dict(model, args.scale, 0))
    model = model.to(dev).eval()
    x = synth.synth_images(1, 640, seed=1).to(dev).half()
    calibrate_cls_bias(model, x, 2000, M, torch)
    plan = model.plan_for(x)
    pred = torch.empty(1, plan.A, 5 + plan.nc, dtype=torch.float32, device=dev)
    res = {}
    for tag, graph in (("hipgraph", True), ("eager", False)):
        for _ in range(args.warmup):
            plan.run_into(x, pred, graph=graph)
            M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
        torch.cuda.synchronize(dev)
        lat_f, lat_t = [], []
        for _ in range(max(args.steps, 200)):
            t0 = time.perf_counter()
            plan.run_into(x, pred, graph=graph)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
            t2 = time.perf_counter()
            lat_f.append(1e3 * (t1 - t0)); lat_t.append(1e3 * (t2 - t0))
        res[tag] = {"forward_ms_p50": round(float(np.percentile(lat_f, 50)), 4), "forward_ms_p99": round(float(np.percentile(lat_f, 99)), 4),
                    "forward_plus_nms_ms_p50": round(float(np.percentile(lat_t, 50)), 4), "forward_plus_nms_ms_p99": round(float(np.percentile(lat_t, 99)), 4)}
    print(json.dumps({"metric": "latency ms MAF-YOLO-%s 640x640 bs=1 infer (hipGraph forward + NMS)" % args.scale,
                      "value": res["hipgraph"]["forward_plus_nms_ms_p50"], "unit": "ms", "n_gpus": 1, "higher_is_better": False,
                      "dtype": "f16", "data": "synthetic", "config": {"workload": "bs=1 3x640x640 fp16, %d launches per forward" % len(plan.ops)},
                      "latency": res}), flush=True)


def train_mode(args, torch, M, dev, rank, world, dist):
    """One step = forward (autocast fp16) + backward + DDP all-reduce + SGD step on a fixed synthetic batch per rank.
    The loss is the device-side ComputeLoss (HIP task-aligned assignment + VFL / GIoU / DFL) on synthetic labels,
    7 boxes per image (the COCO average); --surrogate-loss swaps in a mean over the head outputs (the first rounds' measurement)."""
    from maf_yolo_amd import synth, train_ops
    import torch.nn.functional as F
    if args.torch_convs:                       # A/B: same module tree, convs through F.conv2d (MIOpen)
        train_ops.conv1x1 = lambda x, w, bias=None: F.conv2d(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))
        train_ops.dwconv = lambda x, w: F.conv2d(x, w.to(x.dtype), None, 1, w.shape[-1] // 2, 1, x.shape[1])
    model = M.Model(args.scale)
    model.load_state_dict(synth.synth_state_dict(model, args.scale, 0))
    model = model.to(dev).train()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], gradient_as_bucket_view=True)
    opt = torch.optim.SGD(model.parameters(), lr=0.01 / 64 * args.batch * world, momentum=0.937, nesterov=True, weight_decay=5e-4)
    scaler = torch.amp.GradScaler("cuda")
    B = args.batch
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
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        print(json.dumps({"metric": "train images/sec MAF-YOLO-%s 640x640 bs=%d/GPU DDP (fwd+bwd+all-reduce+SGD, AMP fp16)" % (args.scale, B),
                          "value": round(world * B * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                          "config": {"workload": "MAF-YOLO-%s train-form, %d x 3x640x640 per GPU, %s" % (args.scale, B, "surrogate loss over all head outputs" if args.surrogate_loss else "ComputeLoss (HIP task-aligned assigner + VFL/GIoU/DFL), 7 boxes/image"),
                                     "global_batch": B * world, "parallelism": "ddp%d" % world,
                                     "convs": "torch/MIOpen" if args.torch_convs else "HIP kernels for 1x1 + depth-wise (fwd, dgrad, DW wgrad); 3x3 s2 + BN torch",
                                     "native_launches": dict(train_ops.stats), "final_loss": round(float(loss), 5)}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (BASELINE configs[1]: 32)")
    ap.add_argument("--scale", default="n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--per-op", action="store_true", help="also print the per-op table to stderr")
    ap.add_argument("--lanes", type=int, default=-1, help="engine streams: 0 one stream, 1 heads on side streams, 2 heads + neck side convs (default: the model's setting)")
    ap.add_argument("--fuse", type=int, default=-1, help="1/0: force the fused DepthBottleneckUni kernel on/off (default: the model's setting)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight in the timed serving loop: step i runs on HIP stream i %% S with its own activation arena "
                         "(1 = one stream; the NMS of a batch still overlaps the next forward)")
    ap.add_argument("--tune-file", default=None,
                    help="JSON of autotuned tiles: loaded if it exists (no re-timing: profiler passes run the same kernels as the bench), written after tuning; "
                         "default: profiles/round1_tune.json if present; 'none' = time every layer afresh")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[2]/[3] instead: DDP training step (train-form graph, AMP fp16, SGD), images/s; use with --scale s|m --batch 32|16")
    ap.add_argument("--surrogate-loss", action="store_true", help="with --train: mean over the head outputs instead of ComputeLoss")
    ap.add_argument("--torch-convs", action="store_true", help="with --train: run the 1x1 / depth-wise convs on stock PyTorch-ROCm (MIOpen) for an A/B")
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
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="env://", device_id=dev)

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
    frozen = os.path.join(ROOT, "profiles", "round1_tune.json")
    if args.tune_file is None and os.path.exists(frozen):
        args.tune_file = frozen
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
    # streams that share a hardware queue run their kernels one after the other: take streams that demonstrably overlap (streams.py)
    cs = M.concurrent_streams(dev, S + 1)                      # S forward streams + the NMS stream, on different hardware queues
    streams, nms_stream = (cs[:S] if S > 1 else [torch.cuda.current_stream(dev)]), cs[S]
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
                pred_i = model(x, slot=k)[0]
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
    pipelined(48)
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
            pmc = json.load(open(os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")))
            have = [k for k in gd["inst"] if k in pmc]
            if have:                           # per launch of the template: instantiations weighted by their launches in this forward
                traffic = int(sum(pmc[k]["traffic_bytes"] * groups[k]["n"] for k in have) / sum(groups[k]["n"] for k in have))
                traffic_src = "profiles/round1_pmc_traffic.json: (FETCH_SIZE*2 + WRITE_SIZE)*1024 per launch, gfx950 FETCH_SIZE x2 correction"
        except Exception:
            pass
        insts = []
        for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            a_gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
            insts.append(dict(kernel=k, launches=g["n"], avg_launch_ms=round(g["ms"] / g["n"], 5), share_of_forward=round(g["ms"] / per_op_ms.sum(), 4),
                              achieved_GBs=round(a_gbs, 1), hbm_frac=round(a_gbs / HBM_PEAK_GBS, 4),
                              mfma_frac=round(g["flops"] / (g["ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                              traffic=pmc.get(k, {}).get("traffic_bytes")))
        roofline = dict(bound="hbm", kernel=name + "<...> (%d instantiations)" % len(gd["inst"]), launches_per_forward=gd["n"], avg_launch_ms=round(avg_ms, 5),
                        bytes_per_launch=int(bytes_per_launch), achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                        share_of_forward=round(gd["ms"] / per_op_ms.sum(), 4), top_instantiations=insts,
                        whole_forward=dict(algorithmic_GB=round(tot_bytes / 1e9, 4), GFLOP=round(tot_flops / 1e9, 2),
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
        if world == 1 and not args.no_cpu_baseline:
            from oracle import maf_oracle as O          # the CPU restatement: checker / baseline only
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:                                        # cgroup CPU quota, if any
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    cores = max(1, min(cores, int(int(q) / int(per))))
            except Exception:
                pass
            cores = min(cores, 64)                      # oneDNN convs stop scaling (and start thrashing) beyond this
            torch.set_num_threads(cores)
            dw = O.reparam(sd, args.scale)
            for m_ in model.backbone:          # same calibrated head as the GPU run
                if hasattr(m_, "cls_pred"):
                    key = "backbone.%d.cls_pred" % m_.i
                    dw[key] = (dw[key][0], m_.cls_pred.bias.detach().float().cpu())
            xb = x[:4].float().cpu()
            with torch.no_grad():
                O.non_max_suppression(O.predict(dw, args.scale, xb[:1]).numpy(), conf, iou, multi_label=True)   # warm-up
                n_img, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 12.0 and n_img < 256:
                    p = O.predict(dw, args.scale, xb)
                    O.non_max_suppression(p.numpy(), conf, iou, multi_label=True)
                    n_img += xb.shape[0]
                dt = time.perf_counter() - t0
            cpu = dict(value=round(n_img / dt, 2), unit="images/s", cores=cores, kind="port",
                       sample="%d images (batches of 4, 3x640x640 fp32) through oracle.predict + oracle.non_max_suppression, %.1f s wall, %d torch threads" % (n_img, dt, cores))

        out = {"metric": "images/sec MAF-YOLO-%s 640x640 bs=%d infer (Model.forward + non_max_suppression)" % (args.scale, B),
               "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
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
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
