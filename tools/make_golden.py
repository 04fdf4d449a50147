#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

  python tools/make_golden.py

Imports /root/reference through tools/ref_import.py, loads the synthetic
weights of oracle.maf_oracle.synth_state_dict into the reference's own Model, runs the reference's
train-form forward, its deploy switch (fuse_model / switch_to_deploy / reparameterize), its
deploy-form forward and its non_max_suppression, and records the results as DATA: inputs are
regenerated from seeds (NumPy legacy RandomState), expected outputs are stored.

The only non-reference piece in the loop is the `torchvision.ops.nms` stand-in (torchvision is not
installed; SURVEY.md §8c "parity unpinned"): oracle.maf_oracle.greedy_nms_torch.
Nothing here travels to the GPU box except the .npz files it writes.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_import                      # noqa: E402
from oracle import maf_oracle as O     # noqa: E402
import nms_cases                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def spec_hash(spec):
    return hashlib.sha256(json.dumps([[k, list(s)] for k, s in spec]).encode()).hexdigest()


def wsum(t):
    """Order-sensitive checksum triple of a tensor (float64): sum, sum|.|, sum(x*ramp)."""
    a = t.detach().double().reshape(-1).numpy()
    ramp = (np.arange(a.size) % 97 + 1).astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * ramp).sum()])


def main():
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load(O.greedy_nms_torch)

    for scale in (() if "--nms-only" in sys.argv else ("n", "s", "m")):
        model = ref_import.build(ns, scale)
        ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        spec = O.state_spec(scale)
        assert ref_keys == [(k, tuple(s)) for k, s in spec], "state_dict layout differs from the reference (%s)" % scale
        sd = O.synth_state_dict(scale, seed=0)
        model.load_state_dict(sd, strict=True)
        model.eval()
        deploy = ref_import.to_deploy(ns, model)
        dsd = deploy.state_dict()
        rec = {"spec_hash": np.array(spec_hash(spec)), "n_train_tensors": np.array(len(spec))}
        # (d) train-form -> deploy-form weights: checksum per deploy conv
        names = sorted(k[:-7] for k in dsd if k.endswith(".weight") and k[:-7] + ".bias" in dsd and k.startswith("backbone"))
        rec["deploy_names"] = np.array(names)
        rec["deploy_wsum"] = np.stack([wsum(dsd[n + ".weight"]) for n in names])
        rec["deploy_bsum"] = np.stack([wsum(dsd[n + ".bias"]) for n in names])
        # (b) whole-net predictions, 320x320 B=1 (full) ; train-form rows strided
        x = O.synth_images(1, 320, seed=1)
        with torch.no_grad():
            p_train, feats_train = model(x)
            p_dep, feats_dep = deploy(x)
        rec["pred320_deploy"] = p_dep.numpy()
        rec["pred320_train_rows7"] = p_train[:, ::7].numpy()
        for li, (t, c, r) in enumerate(feats_dep):
            rec["head%d_reg_sum" % li] = wsum(r)
            rec["head%d_cls_sum" % li] = wsum(c)
            rec["head%d_stem_sum" % li] = wsum(t)
        # raw head outputs of the smallest level in full (decode parity input)
        rec["head2_reg"] = feats_dep[2][2].numpy(); rec["head2_cls"] = feats_dep[2][1].numpy()
        # NMS on the reference's own prediction with the reference's eval / infer settings
        for tag, kw in (("eval", dict(conf_thres=0.03, iou_thres=0.65, multi_label=True)),
                        ("infer", dict(conf_thres=0.1, iou_thres=0.45, agnostic=True, max_det=1000)),
                        ("best", dict(conf_thres=0.05, iou_thres=0.45))):
            dets = ns.non_max_suppression(p_dep.clone(), **kw)
            rec["nms320_%s" % tag] = dets[0].numpy()
        if scale == "n":
            # headline shape: 640x640, B=2, every 16th anchor row + float64 column sums
            x = O.synth_images(2, 640, seed=1)
            with torch.no_grad():
                p640, _ = deploy(x)
            rec["pred640_rows16"] = p640[:, ::16].numpy()
            rec["pred640_colsum"] = p640.double().sum(1).numpy()
            dets = ns.non_max_suppression(p640.clone(), conf_thres=0.03, iou_thres=0.65, multi_label=True)
            for bi, d in enumerate(dets):
                rec["nms640_eval_%d" % bi] = d.numpy()
            # (a) per-node outputs of the deploy graph at 64x64 via forward hooks
            taps = {}
            hooks = []
            for m in deploy.backbone:
                hooks.append(m.register_forward_hook(lambda mod, i, o, idx=m.i: taps.__setitem__(idx, o)))
            x = O.synth_images(1, 64, seed=2)
            with torch.no_grad():
                deploy(x)
            for h in hooks:
                h.remove()
            for idx, o in taps.items():
                if isinstance(o, tuple):
                    for j, t in enumerate(o):
                        rec["tap64_%d_%d" % (idx, j)] = t.numpy()
                elif isinstance(o, torch.Tensor):
                    rec["tap64_%d" % idx] = o.numpy()
        np.savez_compressed(os.path.join(OUT, "maf_%s.npz" % scale), **rec)
        print("wrote maf_%s.npz" % scale, {k: getattr(v, "shape", None) for k, v in list(rec.items())[:6]})

    # (c) NMS edge cases through the reference's non_max_suppression
    rec = {}
    for name, (pred, kw) in nms_cases.cases().items():
        dets = ns.non_max_suppression(torch.from_numpy(pred.copy()), **kw)
        rec[name + "__n"] = np.array([d.shape[0] for d in dets])
        for bi, d in enumerate(dets):
            rec["%s__%d" % (name, bi)] = d.numpy()
        # the reference asserts on bad thresholds (nms.py:50-51)
    for bad in (dict(conf_thres=1.5), dict(iou_thres=-0.1)):
        try:
            ns.non_max_suppression(torch.zeros(1, 4, 85), **bad)
            raise SystemExit("reference did not assert")
        except AssertionError:
            pass
    np.savez_compressed(os.path.join(OUT, "nms_cases.npz"), **rec)
    print("wrote nms_cases.npz", {k: v for k, v in rec.items() if k.endswith("__n")})


if __name__ == "__main__":
    main()
