import sys, os, torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth
from maf_yolo_amd.engine import Plan
model = M.Model('n'); model.load_state_dict(synth.synth_state_dict(model, 'n', 0)); model = model.cuda().eval().half()
x = synth.synth_images(32, 640, seed=1).cuda().half()
plan = Plan(model, 32, 640, 640, lib.F16, lib.F16, x.device, fuse=False)  # bottlenecks unfused: every layer shows up
plan.autotune(x, verbose=True)
