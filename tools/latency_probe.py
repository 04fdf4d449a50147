"""Where does a bs = 1 request spend its time?  Host enqueue time of the forward's launch list, device time between the first and the last launch (events),
and the wall time to the host wait — for one request on an idle device, eager and hipGraph.   python tools/latency_probe.py [scale]   (GPU box)"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth

scale = sys.argv[1] if len(sys.argv) > 1 else "m"
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.cuda().eval()
x = synth.synth_images(1, 640, seed=1).cuda().half()
plan = model.plan_for(x)
pred = torch.empty(1, plan.A, 85, dtype=torch.float32, device=x.device)
if len(sys.argv) > 2:                                    # a plan built directly (every fusion on, tiles timed), as tools/latency_per_op.py does
    from maf_yolo_amd.engine import Plan
    plan = Plan(model, 1, 640, 640, lib.F16, lib.F16, x.device)
    plan.autotune(x)
    print("direct plan: %d launches; kernels that differ from plan_for's:" % len(plan.ops))
    p0 = model.plan_for(x)
    a = {p0.op_names[i]: p0.kernel_name(i) for i in range(len(p0.ops))}
    b = {plan.op_names[i]: plan.kernel_name(i) for i in range(len(plan.ops))}
    for k in sorted(set(a) | set(b)):
        if a.get(k) != b.get(k): print("   %-40s %-60s | %s" % (k, a.get(k), b.get(k)))
for graph in (False, True):
    for _ in range(30): plan.run_into(x, pred, graph=graph)
    torch.cuda.synchronize()
    host, wall, dev = [], [], []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        plan.run_into(x, pred, graph=graph)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3); dev.append(e0.elapsed_time(e1))
    print("%-8s host enqueue %.3f ms   device first..last %.3f ms   wall to host wait %.3f ms   (p50 of 200, %d launches)" % ("graph" if graph else "eager", np.median(host), np.median(dev), np.median(wall), len(plan.ops)))
# back to back, no waits in between
for graph in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): plan.run_into(x, pred, graph=graph)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-8s back to back: host %.3f ms / forward, device-bound wall %.3f ms / forward" % ("graph" if graph else "eager", (t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
