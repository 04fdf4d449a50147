#!/usr/bin/env python3
"""Every launch of a recorded training step (maf_yolo_amd/tape.py) timed ON ITS OWN: the step tape of MAF-YOLO-<scale> at batch B is recorded the normal way, then each
record of its forward and backward lists is replayed alone N times between two HIP events on a quiet chip.  What a kernel costs in isolation against what the step's
rocprof trace shows for it (profiles/round5_train_step_kernels.md) tells contention from kernel inefficiency; the weight-gradient launches carry their shapes and bytes.

    python tools/tape_times.py [scale] [batch] [N] > table
"""
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maf_yolo_amd as M                    # noqa: E402
from maf_yolo_amd import lib, synth, tape as tape_mod      # noqa: E402


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "n"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    model = M.Model(scale)
    model.load_state_dict(synth.synth_state_dict(model, scale, 0))
    model = model.to(dev).train()
    ex = M.GradExchange(model)
    x = synth.synth_images(B, 640, seed=1).to(dev)
    g = torch.Generator().manual_seed(100)
    wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
    targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
    crit = M.ComputeLoss(ori_img_size=640, warmup_epoch=0)
    for _ in range(tape_mod.RECORD_AT + 1):
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = model(x)
        loss = crit((feats, cls, reg), targets, 0, 0)[0]
        ex.zero_grad()
        (loss * 1024.0).backward()
    torch.cuda.synchronize()
    (ent,) = model._tapes.values()
    tp = ent[1]
    assert tp is not None and tp.ready, tp and tp.failed
    L = lib._lib
    names = {}
    for nm in lib.EXPORTS:
        i = L.maf_tape_fn_id(nm.encode())
        if i >= 0:
            names[i] = nm
    timer = lib.Timer()
    rows, by = [], collections.defaultdict(lambda: [0, 0.0])
    for which in ("fwd", "bwd"):
        arr, n = tp.arr[which], tp.n[which]
        for i in range(n):
            r = arr[i]
            if r.fn < 0:
                continue
            nm = names.get(r.fn, "?")
            st = tp.harr[r.stream]
            bad = C.c_int32(-1)
            one = (C.c_void_p * len(tp.harr))(*[st] * len(tp.harr))          # every stream index -> the record's own stream: one queue, nothing beside it
            L.maf_tape_run(arr, i, i + 1, one, len(tp.harr), C.byref(bad))
            torch.cuda.synchronize()
            timer.start(st)
            for _ in range(N):
                L.maf_tape_run(arr, i, i + 1, one, len(tp.harr), C.byref(bad))
            timer.stop(st)
            us = 1e3 * timer.elapsed_ms() / N
            note, nbytes = "", 0
            a = r.a
            if nm == "maf_conv_wgrad":
                Bc, Ho, Wo, Hs, Ws, cin, cout, k, s_ = [int(a[j]) for j in range(4, 13)]
                nbytes = (Bc * Hs * Ws * cin + Bc * Ho * Wo * cout) * 2
                note = "x %dx%dx%d -> dy %dx%dx%d k%d s%d" % (Hs, Ws, cin, Ho, Wo, cout, k, s_)
            elif nm == "maf_dw_wgrad":
                Bc, H, W, c, k = [int(a[j]) for j in range(4, 9)]
                nbytes = 2 * Bc * H * W * c * 2
                note = "%dx%dx%d k%d" % (H, W, c, k)
            elif nm == "maf_stem_train":
                Bc, Hi, Wi, co = int(a[2]), int(a[3]), int(a[4]), int(a[7])
                nbytes = (Bc * Hi * Wi * 8 + 2 * Bc * (Hi // 2) * (Wi // 2) * co) * 2
                note = "image %dx%d -> 2 x %d channels" % (Hi, Wi, co)
            elif nm == "maf_dw_wgrad31":
                Bc, H, W, c = [int(a[j]) for j in range(8, 12)]
                nb = 2 if not int(a[4]) else 3
                nbytes = (1 + nb) * Bc * H * W * c * 2
                note = "%dx%dx%d k%s" % (H, W, c, "3+1" if nb == 2 else "3+3+1")
            elif nm == "maf_bn_forward_ex":
                Mp, c = int(a[2]), int(a[3])
                nbytes, note = (3 if not int(a[22]) else 2) * Mp * c * 2, "M %d C %d%s" % (Mp, c, " (statistics ready)" if int(a[22]) else "")
            elif nm == "maf_bn_backward_acc":
                Mp, c = int(a[4]), int(a[5])
                nbytes, note = 5 * Mp * c * 2, "M %d C %d" % (Mp, c)
            elif nm == "maf_bn_stats":
                Mp, c = int(a[2]), int(a[3])
                nbytes, note = Mp * c * 2, "M %d C %d" % (Mp, c)
            elif nm == "maf_bn_sum_forward":
                nb, Mp, c = int(a[2]), int(a[3]), int(a[4])
                nbytes, note = (nb + 1) * Mp * c * 2, "M %d C %d nb %d" % (Mp, c, nb)
            elif nm == "maf_bn_sum_backward":
                nb, Mp, c = int(a[4]), int(a[5]), int(a[6])
                nbytes, note = (2 + 3 * nb) * Mp * c * 2, "M %d C %d nb %d" % (Mp, c, nb)
            elif nm in ("maf_dw_branches", "maf_dw_branches_stats"):
                nb, k0, Bc, H, W, c = [int(a[j]) for j in range(5, 11)]
                nbytes, note = (nb + 1) * Bc * H * W * c * 2, "%dx%dx%d k%d nb %d" % (H, W, c, k0, nb)
            elif nm == "maf_nhwc_sum":
                n_, Mp, c = int(a[2]), int(a[5]), int(a[6])
                nbytes, note = (n_ + 1 + int(a[8])) * Mp * c * 2, "M %d C %d n %d%s" % (Mp, c, n_, " +=" if int(a[8]) else "")
            elif nm == "maf_op_launch":
                op = lib.MafOp.from_address(int(a[0]))
                note = "kind %d %dx%d %d->%d k%d tile (%d,%d,%d)" % (op.kind, op.H, op.W, op.Cin, op.Cout, op.ksize, op.tile_p, op.tile_c, op.tile_k)
                nbytes = op.B * op.H * op.W * (op.Cin + op.Cout) * 2
            rows.append((which, i, r.stream, nm, us, nbytes, note))
            by[(which, r.stream, nm)][0] += 1
            by[(which, r.stream, nm)][1] += us
    print("# every launch of one recorded training step of MAF-YOLO-%s (batch %d), each timed alone (%d repeats, HIP events)" % (scale, B, N))
    print("\n## by entry point\n\n| list | stream | entry point | launches | sum us |\n|---|---|---|---|---|")
    for (which, st, nm), (cnt, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | `%s` | %d | %.0f |" % (which, st, nm, cnt, us))
    tot = collections.defaultdict(float)
    for which, i, st, nm, us, nb, note in rows:
        tot[st] += us
    print("\nsum per stream index (0 main, 1 weight gradients, 2.. lanes), isolated: " + ", ".join("%d: %.2f ms" % (k, v / 1e3) for k, v in sorted(tot.items())))
    print("\n## weight-gradient launches (stream 1)\n\n| entry | shape | us alone | GB/s |\n|---|---|---|---|")
    for which, i, st, nm, us, nb, note in sorted([r_ for r_ in rows if r_[3] in ("maf_conv_wgrad", "maf_dw_wgrad", "maf_dw_wgrad31")], key=lambda r_: -r_[4]):
        print("| `%s` | %s | %.1f | %.0f |" % (nm, note, us, nb / us / 1e3))
    print("\n## every other launch, longest first\n\n| list | # | entry | what | us alone | GB/s (algorithmic bytes of the call) |\n|---|---|---|---|---|---|")
    for which, i, st, nm, us, nb, note in sorted([r_ for r_ in rows if r_[3] not in ("maf_conv_wgrad", "maf_dw_wgrad", "maf_dw_wgrad31")], key=lambda r_: -r_[4]):
        print("| %s | %d | `%s` | %s | %.1f | %s |" % (which, i, nm, note, us, ("%.0f" % (nb / us / 1e3)) if nb else ""))
    ex.close()


if __name__ == "__main__":
    main()
