"""Deterministic synthetic weights / images for benchmarks and smoke runs (no checkpoints or COCO exist
offline).  NumPy legacy RandomState => bit-identical on every box.

Conv weights ~ N(0, g^2/fan_in) with a per-scale gain that keeps |activation| O(1) through the net
(fp16-safe); BN affine/statistics randomised so the deploy switch is non-trivial; the reference
zero-initialises cls_pred/reg_pred (common.py:1307-1323) which would make outputs input-independent,
so those are randomised too.  tests/test_host_logic.py pins this generator to the oracle's own."""
import math

import numpy as np
import torch

_GAIN = {"n": 1.384, "s": 1.438, "m": 1.462}


def synth_state_dict(model, scale="n", seed=0, cls_bias=-5.5):
    rs = np.random.RandomState(seed)
    g_conv = _GAIN.get(scale, 1.4)
    sd = {}
    for key, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            v = np.array(0, dtype=np.int64)
        elif key == "detect.proj":
            v = np.linspace(0, shape[0] - 1, shape[0], dtype=np.float32)
        elif key == "detect.proj_conv.weight":
            v = np.linspace(0, shape[1] - 1, shape[1], dtype=np.float32).reshape(shape)
        elif key.endswith("running_mean"):
            v = (rs.randn(*shape) * 0.1).astype(np.float32)
        elif key.endswith("running_var"):
            v = (0.5 + rs.rand(*shape)).astype(np.float32)
        elif ".bn." in key or "_bn" in key or ".norm." in key:
            v = (0.7 + 0.6 * rs.rand(*shape)).astype(np.float32) if key.endswith(".weight") else (rs.randn(*shape) * 0.1).astype(np.float32)
        elif key.endswith("cls_pred.bias"):
            v = (rs.randn(*shape) * 0.5 + cls_bias).astype(np.float32)
        elif key.endswith("reg_pred.bias"):
            v = (rs.randn(*shape) * 0.5 + 1.0).astype(np.float32)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.2 if "cls_pred." in key else 0.25 if "reg_pred." in key else 0.4 * g_conv if ".dwconv." in key else g_conv
            v = (rs.randn(*shape) * (gain / math.sqrt(fan_in))).astype(np.float32)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v))
    return sd


def synth_images(batch, size, seed=1):
    h, w = (size, size) if isinstance(size, int) else size
    return torch.from_numpy(np.random.RandomState(seed).rand(batch, 3, h, w).astype(np.float32))
