"""Candidate timings of the autotuner for every conv of MAF-YOLO-n at bs 32 (reps=15): which variant wins where, and by how much.
    gpurun -- 'python tools/tune_probe.py > gpurun_out/tune_probe.txt 2>&1'"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth
from maf_yolo_amd.engine import Plan
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.cuda().eval().half()
x = synth.synth_images(bs, 640, seed=1).cuda().half()
plan = Plan(model, bs, 640, 640, lib.F16, lib.F16, x.device, fuse=False)
plan.autotune(x, reps=15, verbose=True)
