"""Every switch of the package in ONE table: attribute, environment override, default, meaning.

`cfg` is read once at import (the environment wins over the defaults); modules take their switches from it (`from .config import cfg`) instead of reading
`os.environ` where they stand.  All switches are A/B or debugging aids — the defaults are the product path, measured in profiles/ — except `hip_lib`
(an instrumented build of the library) and the queue / lane counts, which describe the box.  `python -m maf_yolo_amd.config` prints the table with the
current values.

Model-level settings (`Model.fuse_tail`, `.fuse_bottlenecks`, `.multi_stream`, `.step_tape` ...) stay attributes of the model they belong to; where an
environment switch exists for one of them it only changes what "auto" means and is listed here."""
import os

# attribute: (environment variable, default, meaning).  bool switches: "0" = off, anything else = on; int switches: the number.
SWITCHES = {
    # ---- inference engine (engine.py)
    "dw_stage":          ("MAF_DW_STAGE", True, "tuner: offer dwconv_p2's staged-store form (tile_k + 128) as a candidate"),
    "fuse_mprep":        ("MAF_FUSE_MPREP", True, "MPRep's two branches as one launch where the kernel exists (tuned plans)"),
    "fuse_tail":         ("MAF_FUSE_TAIL", 1, "what Model.fuse_tail = 'auto' means: 0 off, 1 blocks of one bottleneck (default), 3 every instantiation (s / m opt-in)"),
    "split_cat":         ("MAF_SPLIT_CAT", True, "RepHDW behind the fused stem: one dense tensor per concat slot"),
    "mprep_wreg_min":    ("MAF_MPREP_WREG_MIN", 65536, "output pixels from which MPRep takes the register-weight 3x3 kernel"),
    "head_tail_256_max": ("MAF_HEAD_TAIL_256_MAX", 16384, "largest level (B * H * W pixels) whose 256-channel head runs as the fused tail (weights streamed through LDS per 16-pixel unit)"),
    "nms_single_max_batch": ("MAF_NMS_SINGLE_MAX_BATCH", 0, "largest batch the single-launch NMS kernel serves (0: never — the seven-launch path is faster, DESIGN.md 8)"),
    "nms_matrix":        ("MAF_NMS_MATRIX", "auto", "all-pairs NMS: 1 = suppression matrix + row scan, 0 = kept-list scan, auto = kept-list under other work (async), matrix alone up to 32 images"),
    # ---- training step (train_ops.py, tape.py, model.py, solver.py, exchange.py)
    "step_tape":         ("MAF_STEP_TAPE", True, "replay the recorded launch lists of a train step (tape.py) once a batch shape has been seen three times"),
    "stage_native":      ("MAF_STAGE_NATIVE", True, "step tape: the image batch into the static NHWC8 input by one native pass"),
    "train_lanes":       ("MAF_TRAIN_LANES", 2, "lanes (extra streams) a recorded step may put independent branches on"),
    "wgrad_stream":      ("MAF_WGRAD_STREAM", True, "weight gradients on a side stream"),
    "bn_affine_direct":  ("MAF_BN_AFFINE_DIRECT", True, "BatchNorm dgamma / dbeta straight into the exchange's bucket slices"),
    "conv_bn_stats":     ("MAF_CONV_BN_STATS", True, "BatchNorm statistics out of the 1x1 conv's epilogue"),
    "train_tune":        ("MAF_TRAIN_TUNE", True, "time the conv tile variants per shape on first use"),
    "train_tune3":       ("MAF_TRAIN_TUNE3", True, "... the 3x3 stride-2 launches too"),
    "stem_train":        ("MAF_STEM_TRAIN", True, "the image's two RepVGG convs as one direct-conv launch"),
    "dw_wgrad31":        ("MAF_DW_WGRAD31", True, "the 3x3 (+3x3) + 1x1 depth-wise branches' weight gradients as one launch"),
    "dw_branches":       ("MAF_DW_BRANCHES", True, "one launch per direction for the branches of a DilatedReparamBlock"),
    "dw_branch_stats":   ("MAF_DW_BRANCH_STATS", True, "the branches' BatchNorm statistics out of the depth-wise kernel's epilogue"),
    "cat_free":          ("MAF_CAT_FREE", True, "concat nodes without a copy (producers store into their slot)"),
    "bn_sum":            ("MAF_BN_SUM", True, "the branch BatchNorms of a DilatedReparamBlock as one apply pass per direction"),
    "bn_sum_stats":      ("MAF_BN_SUM_STATS", True, "that pass accumulates the statistics of the BatchNorm behind it"),
    "ema_native":        ("MAF_EMA_NATIVE", True, "ModelEMA.update as one launch"),
    "sgd_native":        ("MAF_SGD_NATIVE", True, "the SGD step as one launch"),
    "inf_check_native":  ("MAF_INF_CHECK_NATIVE", True, "GradScaler's inf check as one launch"),
    "exchange_debug":    ("MAF_EXCHANGE_DEBUG", False, "GradExchange keeps the Python stack of every gradient arrival"),
    # ---- the library itself (lib.py)
    "hip_lib":           ("MAF_HIP_LIB", "", "path of another build of libmafyolo_hip.so (make prof / ko / var); empty: the in-tree product library"),
}


# Read by the C library itself (getenv at launch time), for the sweep tools under tools/ only — launch-geometry experiments of the weight-gradient kernels;
# unset in every measured run.  Listed so that the table above plus this one is every environment variable the product looks at.
LIBRARY_SWEEP_HOOKS = {
    "MAF_WGRAD_GX / MAF_WGRAD_R / MAF_WGRAD_TAP_PER_WG": "csrc/wgrad.hip: grid width / replica count / one tap per workgroup of the conv weight gradient (tools/wgrad_sweep.py)",
    "MAF_WGRAD3_GX / MAF_WGRAD3_R / MAF_WGRAD3_PATCH": "csrc/wgrad.hip: the same for the 3x3 patch kernel",
    "MAF_DWWG / MAF_DWWG31 / MAF_DWWG_MFMA / MAF_DWMF_WG": "csrc/train_ops.hip, dw_wgrad_mfma.hip: tiles of the depth-wise weight gradients (tools/dw_wgrad_sweep.py, dw_wgrad31_sweep.py, dw_wgrad_ab.py)",
    "MAF_DWB_TILE / MAF_DWB_TILE_DGRAD": "csrc/dw_branches.hip: tile of the merged depth-wise branches (tools/dwb_sweep.py)",
    "MAF_DGRAD3_ALL_TAPS": "csrc/conv_mfma.hip: A/B of the 3x3 data gradient's tap split",
}


class Config:
    def __init__(self, environ=None):
        environ = os.environ if environ is None else environ
        for attr, (env, default, _) in SWITCHES.items():
            raw = environ.get(env)
            if raw is None:
                val = default
            elif isinstance(default, bool):
                val = raw != "0"
            elif isinstance(default, int):
                val = int(raw)
            else:
                val = raw
            setattr(self, attr, val)

    def overridden(self):
        """{attribute: value} of the switches that differ from their defaults (what a bench line should mention)."""
        return {a: getattr(self, a) for a, (_, d, _m) in SWITCHES.items() if getattr(self, a) != d}

    def table(self):
        rows = ["%-22s %-26s %-10s %s" % ("attribute", "environment", "value", "meaning")]
        for a, (env, d, meaning) in SWITCHES.items():
            v = getattr(self, a)
            rows.append("%-22s %-26s %-10s %s%s" % (a, env, repr(v), meaning, "" if v == d else "   [default %r]" % (d,)))
        return "\n".join(rows)


cfg = Config()

if __name__ == "__main__":
    print(cfg.table())
