"""The tuner's timings per layer and candidate (what it chose between):  python tools/tune_verbose.py [scale] [fused]   (GPU box)"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth
from maf_yolo_amd.engine import Plan
scale = sys.argv[1] if len(sys.argv) > 1 else 'n'
fused = len(sys.argv) > 2
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.cuda().eval().half()
x = synth.synth_images(32, 640, seed=1).cuda().half()
plan = Plan(model, 32, 640, 640, lib.F16, lib.F16, x.device) if fused else Plan(model, 32, 640, 640, lib.F16, lib.F16, x.device, fuse=False)  # bottlenecks unfused: every layer shows up
plan.autotune(x, verbose=True)
