"""ctypes binding of libmafyolo_hip.so (C-ABI: include/mafyolo_hip.h).

The product path has no CPU fallback: if the library is missing this raises, loudly."""
import ctypes as C
import os

from .config import cfg

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = cfg.hip_lib or os.path.join(_HERE, "libmafyolo_hip.so")     # MAF_HIP_LIB: an instrumented build (make prof)

F16, F32, U8 = 0, 1, 2
NMS_FLOAT_THRESHOLD = 1
NMS_PRECOLLECTED = 4         # maf_nms_ex flag: the forward pass (maf_engine_run_filtered) has written the candidate lists of the workspace
NMS_CNT_STRIDE = 64           # MAF_NMS_CNT_STRIDE: ints between the counter lines of two images at the start of the NMS workspace
NMS_MATRIX = 8                # maf_nms_ex flag: the all-pairs path as suppression matrix + scan (rounds 2-5) instead of the kept-list scan (A/B)
NMS_SINGLE_LAUNCH = 2         # maf_nms_ex flag: one launch (collect, then the last workgroup of every image sorts and selects) — the latency path
ACT_NONE, ACT_RELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3
SRC_DIRECT, SRC_UP2, SRC_POOL2, SRC_SUB2, SRC_PAIRS = 0, 1, 2, 3, 4
OP_STEM, OP_CONV1X1, OP_CONV3X3S2, OP_DWCONV, OP_SPPF_POOL, OP_DECODE, OP_BOTTLENECK, OP_CONV1DW, OP_HEADTAIL, OP_STEM2, OP_CONV3X3S2_DGRAD = range(11)


class MafSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("stride", C.c_int32), ("coff", C.c_int32), ("mode", C.c_int32)]


class MafPackDesc(C.Structure):
    """maf_pack_desc_t (include/mafyolo_hip.h): one weight transform of maf_pack_batch."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("total", C.c_int64), ("kind", C.c_int32), ("dtype", C.c_int32),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32), ("transpose", C.c_int32),
                ("CT", C.c_int32), ("steps", C.c_int32), ("Kp", C.c_int32), ("flip", C.c_int32), ("block0", C.c_int32), ("reserved", C.c_int32)]


class MafEmaDesc(C.Structure):
    """maf_ema_desc_t (include/mafyolo_hip.h): one (average, model) tensor pair of maf_ema_update."""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("total", C.c_int64), ("block0", C.c_int32), ("reserved", C.c_int32)]


class MafSgdDesc(C.Structure):
    """maf_sgd_desc_t (include/mafyolo_hip.h): one parameter of maf_sgd_update."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("buf", C.c_void_p), ("total", C.c_int64), ("block0", C.c_int32), ("group", C.c_int32)]


class MafRangeDesc(C.Structure):
    """maf_range_desc_t (include/mafyolo_hip.h): one contiguous fp32 range of maf_nonfinite_check."""
    _fields_ = [("ptr", C.c_void_p), ("total", C.c_int64), ("block0", C.c_int32), ("reserved", C.c_int32)]


class Phase(int):
    """The `phase` argument of a BatchNorm scratch (csrc/bn_act.hip: which half this call accumulates into) as a step tape sees it: a word that alternates
    from replay to replay (tape.py registers a toggle for the argument slot it is passed in)."""


TAPE_MAX_ARGS = 28


class MafTapeRec(C.Structure):
    """maf_tape_rec_t (include/mafyolo_hip.h): one recorded C-ABI call of a step tape."""
    _fields_ = [("fn", C.c_int32), ("stream", C.c_int32), ("a", C.c_uint64 * TAPE_MAX_ARGS)]


class MafTapeToggle(C.Structure):
    _fields_ = [("addr", C.c_void_p), ("mask", C.c_uint64), ("width", C.c_int32), ("reserved", C.c_int32)]


class MafOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("in_dtype", C.c_int32), ("act", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("ksize", C.c_int32), ("nsrc", C.c_int32),
                ("src", MafSrc * 4), ("out", C.c_void_p), ("out_stride", C.c_int32), ("out_coff", C.c_int32),
                ("out_f32", C.c_int32), ("tile_p", C.c_int32), ("tile_c", C.c_int32), ("tile_k", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p),
                ("reg", C.c_void_p * 3), ("lvl_h", C.c_int32 * 3), ("lvl_w", C.c_int32 * 3),
                ("reg_stride", C.c_int32), ("nc", C.c_int32), ("reg_max", C.c_int32), ("lvl_stride", C.c_float * 3),
                ("aux", C.c_void_p * 4), ("lane", C.c_int32), ("n_wait", C.c_int32), ("wait", C.c_int32 * 8),
                ("out_pairs", C.c_int32), ("reserved0", C.c_int32)]


EXPORTS = ["maf_last_error", "maf_version", "maf_op_size", "maf_op_launch", "maf_engine_create", "maf_engine_num_ops",
           "maf_engine_run", "maf_engine_run_filtered", "maf_engine_run_graph", "maf_engine_run_timed", "maf_engine_destroy", "maf_nms_workspace_bytes", "maf_nms", "maf_nms_ex", "maf_nms_debug", "maf_pack_w1x1_bytes", "maf_pack_w1x1", "maf_pack_dw", "maf_pack_batch", "maf_pack_desc_size", "maf_ema_update", "maf_ema_desc_size", "maf_sgd_update", "maf_sgd_desc_size", "maf_nonfinite_check", "maf_range_desc_size", "maf_maxpool_forward", "maf_maxpool_backward", "maf_upsample2x_forward", "maf_upsample2x_backward", "maf_zero", "maf_grad_fold", "maf_add_sub2", "maf_colsum", "maf_dw_wgrad", "maf_dw_wgrad31", "maf_stem_train", "maf_image_to_nhwc8", "maf_bottleneck_record_bytes", "maf_bottleneck_tail_record_bytes", "maf_bottleneck_tail_supported", "maf_conv1dw_record_bytes", "maf_head_tail_record_bytes", "maf_stem2_record_bytes", "maf_conv3s2_lds_record_bytes", "maf_mprep_lds_record_bytes", "maf_mprep_wreg_record_bytes", "maf_conv3s2_wreg_record_bytes", "maf_conv1x1_stats_supported", "maf_coco_rows", "maf_conv1x1_wgrad", "maf_conv_wgrad", "maf_bn_forward", "maf_bn_backward", "maf_bn_backward_acc", "maf_set_deterministic", "maf_dw_branches", "maf_dw_branches_stats", "maf_bn_forward_ex", "maf_bn_replicas", "maf_bn_stats", "maf_bn_sum_forward", "maf_bn_sum_forward_stats", "maf_bn_sum_backward", "maf_tal_targets", "maf_tal_assign", "maf_atss_assign", "maf_loss_partial_rows", "maf_loss_decode", "maf_loss_terms",
           "maf_detect_join", "maf_detect_join_backward", "maf_nhwc_sum", "maf_stream_fork", "maf_stream_join", "maf_tape_fn_id", "maf_tape_fn_nargs", "maf_tape_rec_size", "maf_tape_run", "maf_tape_toggle",
           "maf_stream_create_masked", "maf_stream_destroy",
           "maf_timer_create", "maf_timer_start", "maf_timer_stop", "maf_timer_elapsed_ms", "maf_timer_destroy"]

_lib = None
_recorder = None              # while a step tape records (tape.py): load() hands out a proxy that notes every tape-able call beside making it


class MafError(RuntimeError):
    pass


def load():
    """Load the HIP library (once). Raises MafError if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _recorder is not None:
        return _recorder
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MafError("libmafyolo_hip.so not found at %s — the HIP extension is not built; "
                       "run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.maf_last_error.restype = C.c_char_p
    lib.maf_version.restype = C.c_int
    if not hasattr(lib, "maf_op_size") or lib.maf_op_size() != C.sizeof(MafOp):
        raise MafError("libmafyolo_hip.so was built for a maf_op_t of %s bytes, this binding declares %d: rebuild (__graft_entry__.build())"
                       % (lib.maf_op_size() if hasattr(lib, "maf_op_size") else "?", C.sizeof(MafOp)))
    lib.maf_op_launch.argtypes = [C.POINTER(MafOp), C.c_void_p]
    lib.maf_engine_create.argtypes = [C.POINTER(MafOp), C.c_int32, C.POINTER(C.c_void_p)]
    lib.maf_engine_num_ops.argtypes = [C.c_void_p]
    lib.maf_engine_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.maf_engine_run_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.maf_engine_run_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.maf_engine_destroy.argtypes = [C.c_void_p]
    lib.maf_engine_destroy.restype = None
    lib.maf_engine_run_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    lib.maf_nms_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.maf_nms_workspace_bytes.restype = C.c_int64
    lib.maf_nms.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int32,
                            C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p]
    lib.maf_nms_ex.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_nms_debug.argtypes = [C.POINTER(C.c_uint64)]
    lib.maf_conv1x1_wgrad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_conv_wgrad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] + [C.c_int32] * 10 + [C.c_void_p, C.c_void_p]
    lib.maf_tal_assign.argtypes = ([C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_float] * 3 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_float]
                                   + [C.c_void_p] * 4)
    lib.maf_atss_assign.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 4
    lib.maf_loss_partial_rows.argtypes = [C.c_int32] * 3
    lib.maf_loss_partial_rows.restype = C.c_int64
    lib.maf_loss_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p] * 2
    lib.maf_loss_terms.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_float] * 3 + [C.c_void_p] * 6
    lib.maf_tal_targets.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float] + [C.c_void_p] * 4
    lib.maf_bn_forward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_bn_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_bn_backward_acc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_set_deterministic.argtypes = [C.c_int32]
    lib.maf_dw_branches.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 8 + [C.c_void_p]
    lib.maf_dw_branches_stats.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_bn_forward_ex.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_bn_stats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_bn_sum_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_bn_sum_forward_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_bn_sum_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_bn_replicas.argtypes = [C.c_int32, C.c_int32]
    lib.maf_bn_replicas.restype = C.c_int32
    lib.maf_conv1x1_stats_supported.argtypes = [C.c_int32, C.c_int32]
    lib.maf_conv1x1_stats_supported.restype = C.c_int
    lib.maf_coco_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.maf_conv1dw_record_bytes.argtypes = [C.c_int32] * 2
    lib.maf_conv1dw_record_bytes.restype = C.c_int64
    lib.maf_head_tail_record_bytes.argtypes = [C.c_int32]
    lib.maf_head_tail_record_bytes.restype = C.c_int64
    lib.maf_stem2_record_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.maf_stem2_record_bytes.restype = C.c_int64
    lib.maf_conv3s2_lds_record_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.maf_conv3s2_lds_record_bytes.restype = C.c_int64
    lib.maf_mprep_lds_record_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.maf_mprep_lds_record_bytes.restype = C.c_int64
    lib.maf_mprep_wreg_record_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.maf_mprep_wreg_record_bytes.restype = C.c_int64
    lib.maf_conv3s2_wreg_record_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.maf_conv3s2_wreg_record_bytes.restype = C.c_int64
    lib.maf_bottleneck_record_bytes.argtypes = [C.c_int32] * 3
    lib.maf_bottleneck_record_bytes.restype = C.c_int64
    lib.maf_bottleneck_tail_record_bytes.argtypes = [C.c_int32] * 3
    lib.maf_bottleneck_tail_record_bytes.restype = C.c_int64
    lib.maf_bottleneck_tail_supported.argtypes = [C.c_int32] * 4
    lib.maf_pack_w1x1_bytes.argtypes = [C.c_int32] * 5
    lib.maf_pack_w1x1_bytes.restype = C.c_int64
    lib.maf_pack_w1x1.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_pack_dw.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_pack_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_pack_desc_size.argtypes = []
    lib.maf_ema_update.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p]
    lib.maf_ema_desc_size.argtypes = []
    lib.maf_sgd_update.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.maf_sgd_desc_size.argtypes = []
    lib.maf_nonfinite_check.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_range_desc_size.argtypes = []
    lib.maf_zero.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.maf_grad_fold.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_upsample2x_forward.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_upsample2x_backward.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_add_sub2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    lib.maf_colsum.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_maxpool_forward.argtypes = [C.c_void_p] + [C.c_int32] * 9 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_maxpool_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_pack_desc_size.restype = C.c_int32
    lib.maf_ema_desc_size.restype = C.c_int32
    lib.maf_sgd_desc_size.restype = C.c_int32
    lib.maf_range_desc_size.restype = C.c_int32
    lib.maf_dw_wgrad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.maf_image_to_nhwc8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.maf_stem_train.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.maf_dw_wgrad31.argtypes = [C.c_void_p, C.c_int32] * 4 + [C.c_int32] * 5 + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p]
    lib.maf_detect_join.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 3
    lib.maf_detect_join_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_nhwc_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.maf_stream_fork.argtypes = [C.c_void_p, C.c_void_p]
    lib.maf_stream_join.argtypes = [C.c_void_p, C.c_void_p]
    lib.maf_stream_create_masked.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.maf_stream_destroy.argtypes = [C.c_void_p]
    lib.maf_tape_fn_id.argtypes = [C.c_char_p]
    lib.maf_tape_fn_id.restype = C.c_int32
    lib.maf_tape_fn_nargs.argtypes = [C.c_int32]
    lib.maf_tape_fn_nargs.restype = C.c_int32
    lib.maf_tape_rec_size.restype = C.c_int32
    lib.maf_tape_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.maf_tape_toggle.argtypes = [C.c_void_p, C.c_int32]
    if lib.maf_tape_rec_size() != C.sizeof(MafTapeRec):
        raise MafError("libmafyolo_hip.so was built for a maf_tape_rec_t of %d bytes, this binding declares %d: rebuild" % (lib.maf_tape_rec_size(), C.sizeof(MafTapeRec)))
    lib.maf_timer_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.maf_timer_start.argtypes = [C.c_void_p, C.c_void_p]
    lib.maf_timer_stop.argtypes = [C.c_void_p, C.c_void_p]
    lib.maf_timer_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.maf_timer_destroy.argtypes = [C.c_void_p]
    lib.maf_timer_destroy.restype = None
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MafError("libmafyolo_hip: %s (code %d)" % (load().maf_last_error().decode(), rc))


class Timer:
    """HIP events recorded on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(load().maf_timer_create(C.byref(self._h)))

    def start(self, stream):
        check(load().maf_timer_start(self._h, stream))

    def stop(self, stream):
        check(load().maf_timer_stop(self._h, stream))

    def elapsed_ms(self):
        ms = C.c_float()
        check(load().maf_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._h:
                load().maf_timer_destroy(self._h)
        except Exception:
            pass
