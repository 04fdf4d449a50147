"""Fixture for the checkpoint bridge: a checkpoint written the way the reference's trainer writes it
(yolov6/core/engine.py:195-201: pickled fp16 modules under 'model' and 'ema') for a down-scaled MAF-YOLO graph, plus the
outputs of the reference model on a seeded image.  Run in the build container:

    python tools/make_golden_ckpt.py     ->  tests/golden/ref_ckpt_tiny.pt, tests/golden/ref_ckpt_tiny.npz

The .pt holds tensors and class *names* of the reference (what any of its checkpoints holds), no source."""
import copy
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402
from oracle import maf_oracle as O  # noqa: E402


def tiny_yaml():
    with open(os.path.join(ref_import.REF, "configs/yaml/MAF-YOLO-n.yaml")) as f:
        d = yaml.safe_load(f)
    d["width_multiple"] = 0.125
    lit = {48: 16, 96: 32, 192: 64, 384: 64, 64: 16, 128: 32}
    for sec in ("backbone", "neck"):
        for row in d[sec]:
            if row[2] in ("RepHDW", "ConvWrapper"):
                row[3][0] = lit[row[3][0]]
    return d


def main():
    ns = ref_import.load(lambda b, s, t: torch.zeros(0, dtype=torch.long))
    d = tiny_yaml()
    with tempfile.TemporaryDirectory() as tmp:
        yp = os.path.join(tmp, "tiny.yaml")
        with open(yp, "w") as f:
            yaml.safe_dump(d, f)
        cfg = ref_import.AttrDict(model=dict(type="tiny", build_type="yaml", yaml_file=yp, pretrained=None,
                                             head=dict(type="EffiDeHead", num_layers=3, anchors=1, strides=[8, 16, 32], iou_type="giou", use_dfl=True, reg_max=16)))
        with open(os.devnull, "w") as dn:
            import contextlib
            with contextlib.redirect_stdout(dn):
                model = ns.Model(cfg, channels=3, num_classes=80, anchors=1).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() > 1 and "proj" not in n_:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1.0, (p[0].numel()) ** 0.5)))
        for n_, b in model.named_buffers():
            if n_.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
            if n_.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
    ema = copy.deepcopy(model)
    with torch.no_grad():
        for n_, p in ema.named_parameters():
            if p.dim() > 1 and "proj" not in n_:                      # proj / proj_conv are the fixed DFL bins (yolo.py:327-330)
                p.mul_(0.5)                                           # EMA differs from the live model: the loader must pick 'ema'
    ckpt = {"model": copy.deepcopy(model).half(), "ema": copy.deepcopy(ema).half(), "updates": 17,
            "optimizer": None, "epoch": 3}
    out = os.path.join(ROOT, "tests", "golden")
    torch.save(ckpt, os.path.join(out, "ref_ckpt_tiny.pt"))
    ref = copy.deepcopy(ema).half().float().eval()                    # what load_checkpoint(...).float() yields (checkpoint.py:87)
    x = O.synth_images(1, 64, 3)
    with torch.no_grad():
        y = ref(x)[0]
    sd = ref.state_dict()
    np.savez_compressed(os.path.join(out, "ref_ckpt_tiny.npz"), pred=y.numpy(), n_keys=np.asarray(len(sd)),
                        key_hash=np.asarray([hash_keys(sd)]), checksum=np.asarray([float(sum(v.double().sum() for v in sd.values()))]))
    print("keys", len(sd), "pred", tuple(y.shape), "file bytes", os.path.getsize(os.path.join(out, "ref_ckpt_tiny.pt")))


def hash_keys(sd):
    import hashlib
    return int(hashlib.sha1("\n".join("%s %s" % (k, tuple(v.shape)) for k, v in sd.items()).encode()).hexdigest()[:12], 16)


if __name__ == "__main__":
    main()
