"""Per-launch times of the bs = 1 forward (BASELINE configs[4]: m at 640 x 640), each launch timed alone on a quiet stream:
   python tools/latency_per_op.py [scale]   (GPU box)   — where do the 1.9 ms of 137 dependent launches go?"""
import sys, os, ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth
from maf_yolo_amd.engine import Plan

scale = sys.argv[1] if len(sys.argv) > 1 else "m"
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.cuda().eval().half()
x = synth.synth_images(1, 640, seed=1).cuda().half()
plan = Plan(model, 1, 640, 640, lib.F16, lib.F16, x.device)
plan.autotune(x)
pred = torch.empty(1, plan.A, 85, dtype=torch.float32, device=x.device)
plan.run_into(x, pred)
torch.cuda.synchronize()
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
timer = lib.Timer()
n = len(plan.ops)
ts = np.zeros((n, 30))
for rep in range(30):
    ts[:, rep] = np.array(plan.run_timed(x, pred)) * 1e3
med = np.median(ts, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20): plan.run_into(x, pred)
torch.cuda.synchronize(); e0.record()
for _ in range(200): plan.run_into(x, pred)
e1.record(); torch.cuda.synchronize()
print("forward %.1f us back-to-back; sum of isolated launches %.1f us over %d launches" % (e0.elapsed_time(e1) * 5, med.sum(), n))
for i in np.argsort(-med)[:45]:
    o = plan.ops[i]
    print("%-36s %-58s %7.1f us  %dx%d %d->%d" % (plan.op_names[i], plan.kernel_name(i), med[i], o.H, o.W, o.Cin, o.Cout))
