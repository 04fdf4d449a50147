"""Time chosen (tile_p, tile_c, tile_k) variants of chosen conv layers of MAF-YOLO-n (bs 32, 640^2) — with the shipped library or a knock-out build
(MAF_HIP_LIB=maf-yolo_amd/libmafyolo_ko<bits>.so, csrc/Makefile `make ko KO=<bits>`): what does each piece of the kernel cost?
    python tools/conv_probe.py backbone.20.conv1:2,8,1:1,4,2 backbone.22.conv2:2,8,1"""
import sys, os, ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import maf_yolo_amd as M
from maf_yolo_amd import lib, synth, pack
from maf_yolo_amd.engine import Plan

specs = []
for a in sys.argv[1:]:
    parts = a.split(":")
    specs.append((parts[0], [tuple(int(v) for v in p.split(",")) for p in parts[1:]]))
scale = os.environ.get("MAF_PROBE_SCALE", "n")                     # MAF_PROBE_SCALE=m: the layers of another scale
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.cuda().eval().half()
x = synth.synth_images(32, 640, seed=1).cuda().half()
plan = Plan(model, 32, 640, 640, lib.F16, lib.F16, x.device, fuse=False)
pred = torch.empty(32, plan.A, 85, dtype=torch.float32, device=x.device)
plan.run_into(x, pred)
torch.cuda.synchronize()
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
timer = lib.Timer()
tag = os.path.basename(os.environ.get("MAF_HIP_LIB", "shipped"))
for name, cands in specs:
    i = plan.op_names.index(name)
    o, r = plan.ops[i], plan._ops[i]
    ops = []
    if o.kind == lib.OP_DWCONV:                      # depth-wise: (tile_p, tile_c, tile_k) as the tuner sets them (-2, columns, rows * 256 + channels = csrc/dwconv_dot2.hip)
        for pt, ct, tk in cands:
            op = lib.MafOp.from_buffer_copy(o)
            op.tile_p, op.tile_c, op.tile_k = pt, ct, tk
            keep = None
            if pt == -4:                                # pixel-pair input (csrc/dwconv_p2.hip): the NHWC content of the buffer read as pairs times the same
                kk = o.ksize * o.ksize
                wkc = plan.weights[r["w"]:r["w"] + kk * o.Cout * 2].view(torch.float16).reshape(kk, o.Cout)
                keep = pack.pack_dw_pairs(wkc.t().reshape(o.Cout, 1, o.ksize, o.ksize).float().cpu()).cuda()
                op.src[0].mode, op.aux[1] = lib.SRC_PAIRS, keep.data_ptr()
            ops.append((pt, ct, tk, op, keep))
        cands = []
    else:
        w, b, srcC = r["raw"]
    for pt, ct, tk in cands:
        wp = (pack.pack_conv3x3_lds(w, b) if tk == 6 else pack.pack_conv3x3_wreg(w, b) if tk == 7 else pack.pack_conv1x1(w, srcC, ct, lib.F16) if o.kind == lib.OP_CONV1X1 else pack.pack_conv3x3(w, ct, lib.F16)).cuda()
        bp = pack.pack_bias(b, ct if tk != 6 else 4).cuda()
        op = lib.MafOp.from_buffer_copy(o)
        op.tile_p, op.tile_c, op.tile_k, op.w, op.bias = pt, ct, tk, wp.data_ptr(), bp.data_ptr()
        keep = [wp, bp]
        if "twin" in r:
            w2, b2 = r["twin"]["raw"]
            wp2 = pack.pack_conv3x3(w2, ct, lib.F16).cuda(); bp2 = pack.pack_bias(b2, ct).cuda()
            op.aux[1], op.aux[2] = wp2.data_ptr(), bp2.data_ptr()
            keep += [wp2, bp2]
        ops.append((pt, ct, tk, op, keep))
    ts = {c[:3]: [] for c in ops}
    for c in ops:
        lib.check(L.maf_op_launch(C.byref(c[3]), st))
    for rep in range(25):
        for c in ops:
            timer.start(st); lib.check(L.maf_op_launch(C.byref(c[3]), st)); timer.stop(st)
            ts[c[:3]].append(timer.elapsed_ms() * 1e3)
    by = plan.algorithmic_bytes(i)
    print("%-18s %-38s %6.1f MB  " % (tag, name, by / 1e6) + "  ".join("(%d,%d,%d) min %.1f med %.1f us" % (k + (min(v), float(np.median(v)))) for k, v in ts.items()), flush=True)
