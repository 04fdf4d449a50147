/*
 * mafyolo_hip.h — C-ABI of libmafyolo_hip.so: the MI355X (gfx950) native hot path of MAF-YOLO.
 *
 * The reference (yang-0201/MAF-YOLO) is pure PyTorch: it has no FFI layer, its "operator API" for
 * this path is the Python surface Model.forward() / non_max_suppression().  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to run that path natively:
 * plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless noted; all
 * launches go to the hipStream_t passed in (pass the framework's current stream); nothing here
 * allocates device memory, synchronises the device or keeps references after return, except
 * maf_engine_* which keeps a host-side copy of the op list.
 *
 * Every entry point returns 0 on success or a negative MAF_E_* code; maf_last_error() returns a
 * thread-local message.  Reference lines cited are relative to /root/reference.
 *
 * Activation layout is NHWC ("pixels x channels"), element type f16 or f32 (MAF_F16 / MAF_F32),
 * fp32 accumulation everywhere.  A tensor view is (ptr, stride, coff): channel c of pixel m lives
 * at ptr[m*stride + coff + c] — producers write straight into channel slices of wider buffers, so
 * torch.cat / Tensor.split / nn.Upsample / MaxPool2d(2) of the reference (common.py:154,670,940,944;
 * MAF-YOLO-n.yaml:21,26) never materialise.
 */
#ifndef MAFYOLO_HIP_H
#define MAFYOLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* maf_stream_t;              /* hipStream_t */

enum { MAF_F16 = 0, MAF_F32 = 1, MAF_U8 = 2 };
enum { MAF_ACT_NONE = 0, MAF_ACT_RELU = 1, MAF_ACT_SILU = 2, MAF_ACT_SIGMOID = 3 };
enum { MAF_SRC_DIRECT = 0,               /* source has the op's H x W grid                              */
       MAF_SRC_UP2 = 1,                  /* source is H/2 x W/2, read through nearest x2 upsample       */
       MAF_SRC_POOL2 = 2,                /* source is 2H x 2W, read through MaxPool2d(2,2)              */
       MAF_SRC_SUB2 = 3,                 /* source is 2H x 2W, pixel (2y, 2x) is read: a 1x1 conv with stride 2 (RepVGGBlock.rbr_1x1)   */
       MAF_SRC_PAIRS = 4 };              /* source has the op's grid, stored as PIXEL PAIRS: [B][H][W/2][stride][2] halfs, channel c of pixel x at
                                            ((b H + y) W + (x & ~1)) stride + 2 (coff + c) + (x & 1) — what a CONV1X1 with out_pairs = 1 writes and a
                                            DWCONV with tile_p = -4 reads (csrc/dwconv_p2.hip); fp16, even W, the buffer holds nothing else              */
enum { MAF_OP_STEM = 0,                  /* RepVGGBlock L0 deploy form: 3x3 s2 conv on the NCHW image   */
       MAF_OP_CONV1X1 = 1,               /* Conv 1x1 (+bias+act) over up to 4 concatenated sources      */
       MAF_OP_CONV3X3S2 = 2,             /* RepVGGBlock / ConvWrapper 3x3 stride 2 pad 1 (+bias+act)    */
       MAF_OP_DWCONV = 3,                /* merged DilatedReparamBlock: depth-wise k x k s1 (+bias+act) */
       MAF_OP_SPPF_POOL = 4,             /* three chained MaxPool2d(5,1,2) into concat slices           */
       MAF_OP_DECODE = 5,                /* Detect_yaml eval branch: DFL decode -> [B,A,5+nc] fp32      */
       MAF_OP_BOTTLENECK = 6,            /* fused DepthBottleneckUni: 1x1 -> depth-wise k x k -> 1x1    */
       MAF_OP_CONV1DW = 7,               /* first half of a DepthBottleneckUni: 1x1 (c -> 3c) + SiLU -> depth-wise k x k + SiLU */
       MAF_OP_HEADTAIL = 8,              /* one detection level: {cls,reg}_conv_s -> {cls,reg}_pred -> sigmoid / DFL decode into the prediction rows */
       MAF_OP_STEM2 = 9,                 /* backbone.0 + backbone.1: image -> 1/4-resolution map, the 1/2-resolution tensor stays in LDS */
       MAF_OP_CONV3X3S2_DGRAD = 10 };    /* training: data gradient of a 3x3 stride-2 pad-1 conv (gather form, no atomics)              */
enum { MAF_E_ARG = -1, MAF_E_UNSUPPORTED = -2, MAF_E_HIP = -3 };

typedef struct {
    const void* ptr;
    int32_t C;        /* channels consumed from this source                         */
    int32_t stride;   /* elements per pixel of the underlying buffer                */
    int32_t coff;     /* first channel                                              */
    int32_t mode;     /* MAF_SRC_*                                                  */
} maf_src_t;

/*
 * One launch of the plan.  Fields not used by a kind are ignored.
 *
 * MAF_OP_STEM       replaces backbone.0.rbr_reparam + ReLU (yolov6/layers/common.py:216-217).
 *                   src[0].ptr = image [B,3,2H,2W] NCHW of type in_dtype (f16/f32, or u8 scaled by
 *                   1/255 = evaler.py:161-163 folded in); w = fp32 [27][Cout] (k = (c*3+ky)*3+kx).
 * MAF_OP_CONV1X1    replaces Conv.forward_fuse (common.py:49-50) and the plain nn.Conv2d preds
 *                   (common.py:1331,1335).  w = MFMA-fragment-packed (maf_pack_* in pack.py),
 *                   K = concat of the sources' channels in order (= torch.cat order).
 *                   tile_k = 5 (fp16, 2 <= K / 32 <= 24, K / 32 * tile_c <= 160): persistent workgroups with the channel tile's weight fragments resident
 *                   in LDS (csrc/conv_stream_lds.inc.h); tile_p = 1: four waves per workgroup, 2 (64 <= K / 32 * tile_c, K / 32 >= 8, tile_c in {4, 6, 8}):
 *                   eight waves behind the same LDS copy (csrc/conv_stream_lds_w8.hip) — same result bit for bit.
 *                   tile_k = 2 / 8 (fp16; CONV1X1 with direct / up-sampled sources and CONV3X3S2; tile_c in {4, 6, 8}, tile_p in {1, 2}, tile_k = 2 also
 *                   (4, 4)): the four waves of a workgroup share each k-step's weight fragments through LDS — 2: register -> LDS copy and a barrier per
 *                   k-step; 8: the fragments arrive by DMA (buffer load with LDS destination) in stages of two k-steps through a ring of four slots, three
 *                   stages ahead, one barrier per stage (csrc/conv_mfma_dma.hip: the K-heavy layers of the 40 x 40 / 20 x 20 maps of s and m).
 * MAF_OP_CONV3X3S2  replaces rbr_reparam / ConvWrapper.block.conv stride-2 convs.  src[0] is the
 *                   2H x 2W input (Hin, Win below); w packed per tap.
 *                   Twin launch (CONV1X1 with one direct / pooled source, CONV3X3S2; tile_k 0 / 1 / 2 / 4 / 8): aux[0] != NULL runs a SECOND conv of
 *                   identical shape, strides and tiles as blockIdx.y = 1 of the same grid — aux = {src, w, bias, out} of the second conv
 *                   (the two side ConvWrappers of a MAFPN level, configs/yaml/MAF-YOLO-n.yaml nodes 23 / 24 and 27 / 28).
 *                   tile_k = 6 (fp16; (Cin, Cout) = (48, 48), (48, 64), (64, 64)): persistent workgroups with all weight fragments and the
 *                   input patch of a 4 x 16 output tile in LDS (csrc/conv3s2_lds.hip); w = record of maf_conv3s2_lds_record_bytes(Cin, Cout)
 *                   bytes (maf-yolo_amd/pack.py:pack_conv3x3_lds: fragments + bias), bias unused, tile_c = workgroups / 64 (0 = 256).
 *                   With nc = C1 > 0 ((Cin, Cout, C1) = (48, 48, 48) or (64, 64, 64), even Hin / Win) the launch is a whole MPRep (common.py:776-792,
 *                   cat(conv1(MaxPool2d(2, 2)(x)), conv2(x))): the 2 x 2 windows of a tile's pixels lie inside its staged patch, so the pooled
 *                   1x1 + SiLU branch is taken from LDS and its C1 channels go to `out` at channel offset reg_stride (beside the conv's
 *                   out_coff .. + Cout); w = record of maf_mprep_lds_record_bytes(Cin, Cout, C1) bytes (pack.py:pack_mprep_lds).
 *                   tile_k = 7 (fp16; (Cin, Cout) = (128, 128), (96, 96), (96, 64), (64, 64)): one workgroup per CU whose waves keep ALL weight
 *                   fragments in registers, input patches of 4 x 8 output tiles by DMA, double-buffered (csrc/conv3s2_wreg.hip); w = record of
 *                   maf_conv3s2_wreg_record_bytes(Cin, Cout) bytes (pack.py:pack_conv3x3_wreg), bias unused, tile_c = workgroups per conv / 32
 *                   (0 = 256 / convs); a twin launch passes aux = {src, record, -, out} of the second conv.  With nc = C1 = 96 ((96, 96), no
 *                   twin, even Hin / Win, reg_stride + nc = out_coff) the launch is a whole MPRep as for tile_k = 6: the 2 x 2 window of an output
 *                   pixel is taps (1,1) (1,2) (2,1) (2,2) of its 3 x 3 window, so the pooled branch's operand is the maximum of four fragments the
 *                   conv reads anyway; w = record of maf_mprep_wreg_record_bytes(Cin, Cout, C1) bytes (pack.py:pack_mprep_wreg).
 * MAF_OP_CONV3X3S2_DGRAD  backward of the above w.r.t. its input (autograd of common.py:219-224 / :44-47 in Trainer.train_in_steps,
 *                   yolov6/core/engine.py:164): src[0] = dY [B,Hin,Win,Cin] (Cin = the forward conv's OUTPUT channels, Hin x Win its output
 *                   grid), out = dX [B,H,W,Cout] (Cout = the forward conv's input channels, H x W its input grid); w = the forward weight
 *                   with the channel axes swapped, packed per tap like CONV3X3S2's; bias = zeros; act = NONE; tile_p in {1,2}, tile_c in {2,4,8}.
 * MAF_OP_DWCONV     replaces DilatedReparamBlock.lk_origin after merge (common.py:3025,3033-3051).
 *                   w = [k*k][C] of the activation dtype.  tile_p / tile_c / tile_k optionally fix the workgroup tile
 *                   (rows, cols, channels per block); 0 = built-in cost model.  tile_p = -1 (fp16) selects the matrix-core
 *                   variant (csrc/dwconv_mfma.hip), whose operand is aux[0] = the Toeplitz table [C/32][8][k][parts][16][8] f16
 *                   (maf-yolo_amd/pack.py:pack_dw_toeplitz).  The other fp16 variants (same result, same operands unless noted):
 *                   tile_p = -2  v_dot2_f32_f16 over tap pairs of a pair-interleaved halo tile (csrc/dwconv_dot2.hip): tile_c = tile columns
 *                                (multiple of 8), tile_k = tile rows * 256 + channels per block (multiple of 8, <= 64);
 *                   tile_p = -3  a wave per 8-channel group, halo plane by DMA, weights as scalar operands of v_fma_mix_f32 (csrc/dwconv_sw.hip):
 *                                tile_c = tile columns (multiple of 4), tile_k = tile rows * 256 + waves per workgroup (1..8);
 *                   tile_p = -4  the same work split on v_dot2c_f32_f16 with scalar weight PAIRS over an input stored as pixel pairs
 *                                (csrc/dwconv_p2.hip): src[0].mode = MAF_SRC_PAIRS (written by the CONV1X1 in front with out_pairs = 1), even W,
 *                                aux[1] = weight pairs [C/8][k][2][2][(k+1)/2][4] dwords (pack.py:pack_dw_pairs); tile_c / tile_k as for -3.
 *                   Cout = 2 Cin runs two filters per input channel (output channel c reads input channel c mod Cin: the head's cls / reg pair).
 * MAF_OP_SPPF_POOL  replaces SPPF.m x3 (common.py:121-129): src[0] = x (slice 0 of the 4c_ buffer),
 *                   out/out_coff = slice 1; slices 2 and 3 follow at +C each.
 * MAF_OP_BOTTLENECK replaces one DepthBottleneckUni in deploy form (common.py:918-927: 1x1 c->3c + SiLU, depth-wise k x k + SiLU,
 *                   1x1 3c->c + SiLU) in one launch.  w = tile_k block records of maf_bottleneck_record_bytes() bytes, one per
 *                   32 mid channels: W1 fragments [2][S1][64][8] f16 | b1 [32] f32 | Toeplitz table of the depth-wise filter
 *                   [8][k][parts][16][8] f16 | W2 fragments [CT2][64][8] f16 | bdw [32] f32 (maf-yolo_amd/pack.py:
 *                   pack_bottleneck); bias = b2 fp32 padded to 16*CT2.  fp16 only, Cin = Cout = c <= 64, act = SiLU.
 *                   nc = C3 > 0 (where maf_bottleneck_tail_supported(k, c, nsrc, C3)): the launch also applies the 1x1 conv + SiLU that closes the
 *                   surrounding RepHDW block (common.py:944-946: conv2(torch.cat(y, 1)), y = [x1, x2, block outputs]) to its tile before anything is stored —
 *                   the LAST bottleneck of a block: src[1 .. nsrc-1] = the concat slots in front of this bottleneck's input (c channels each, direct),
 *                   src[0] = its input, then its own result from registers; aux[0] = record of maf_bottleneck_tail_record_bytes(c, nsrc, C3) bytes
 *                   (maf-yolo_amd/pack.py:pack_bottleneck_tail); out / out_stride / out_coff = the CLOSING conv's output (C3 channels) — the
 *                   bottleneck's own output is never written.
 * MAF_OP_CONV1DW    conv1 -> conv2 -> act of DepthBottleneckUni (common.py:905-909) for any width: Cin = c, Cout = 3c, ksize = k;
 *                   w = ceil(3c/32) block records of maf_conv1dw_record_bytes(k, c) bytes (W1 fragments [2][S1][64][8] f16 | b1 [32]
 *                   f32 | Toeplitz table [8][k][parts][16][8] f16 | bdw [32] f32; maf-yolo_amd/pack.py:pack_conv1dw); fp16, act = SiLU.
 * MAF_OP_STEM2      replaces backbone.0 AND backbone.1 (two RepVGGBlocks in deploy form: 3x3 s2 conv + ReLU each, common.py:216-217) in one
 *                   launch.  src[0].ptr = image [B,3,Hin,Win] NCHW of in_dtype (u8: /255 folded); H, W = the 1/4-resolution grid; Cin = 3,
 *                   ksize = C0 (channels of backbone.0), Cout = C1: (24, 48) or (32, 64); nc = C3: 0, or C1 to also apply the 1x1 conv + SiLU
 *                   that opens the next RepHDW block (backbone.2.conv1, common.py:898-946) before anything is written — out then receives
 *                   C3 channels; w = record of maf_stem2_record_bytes(C0, C1, C3) bytes (maf-yolo_amd/pack.py:pack_stem2); fp16 engine only;
 *                   tile_p = tile rows (0 / 8, or 4), tile_k = workgroups (0 = default, persistent).  aux[0] != NULL (needs nc = C1): RepHDW
 *                   splits that tensor in two (x.split((c_, c_), 1), common.py:940) — channels C3/2.. then go to aux[0], a tensor of its own with
 *                   pixel stride reg_stride (elements, a multiple of 8), and out receives channels 0..C3/2-1: a reader of ONE half then
 *                   fetches whole cache lines of what it uses instead of every line of an interleaved buffer.
 * MAF_OP_HEADTAIL   replaces, for ONE level, cls_conv_s + cls_pred + sigmoid and reg_conv_s + reg_pred (Head_DepthUni, common.py:1288-1336:
 *                   Conv.forward_fuse, nn.Conv2d) and that level's share of the Detect_yaml eval branch (yolo.py:355-396) in one launch.
 *                   src[0] / src[1] = inputs of cls_conv_s / reg_conv_s (C = Cin = head width: 64, 128, 192 — weights LDS-resident — or 256, 384 — weights streamed from L2); w / aux[0] = weight
 *                   records of the cls / reg branch, maf_head_tail_record_bytes(Cin) bytes each (maf-yolo_amd/pack.py:pack_head_tail);
 *                   out = pred fp32 [B, A, 85] (supplied per run like DECODE's); Hin = first anchor of the level, Win = A (anchors
 *                   per image), lvl_stride[0] = stride; nc = 80, reg_max = 16; fp16 only.  tile_k = units per wave (0 = auto).
 * MAF_OP_DECODE     replaces Detect_yaml.forward eval branch (yolov6/models/yolo.py:355-396).
 *                   src[l] (l<3): ptr = cls fp32 [B,HW_l,src[l].stride] (row stride >= nc; 0 = nc: the pred conv pads nc to a multiple
 *                   of 4 channels so that any class count runs), and reg[l] = fp32 [B,HW_l,reg_stride] (both NULL: the level's rows are
 *                   written by a MAF_OP_HEADTAIL, lvl_h / lvl_w still give its size);
 *                   out = pred fp32 [B, A, 5+nc].
 */
typedef struct {
    int32_t kind, dtype, in_dtype, act;
    int32_t B, H, W;             /* output grid per image                                       */
    int32_t Hin, Win;            /* input grid (STEM, CONV3X3S2)                                */
    int32_t Cin, Cout, ksize;
    int32_t nsrc;
    maf_src_t src[4];
    void* out;
    int32_t out_stride, out_coff;
    int32_t out_f32;             /* CONV1X1: store fp32 regardless of dtype (cls_pred/reg_pred) */
    int32_t tile_p, tile_c;      /* MFMA tile: 16*tile_p pixels x 16*tile_c channels per wave   */
    int32_t tile_k;              /* 0/1: each wave reduces all of K; 4: the 4 waves of a workgroup split K (tile_p = 1); 2 / 8: weight fragments shared through LDS */
    const void* w;
    const float* bias;
    /* DECODE only */
    const void* reg[3];
    int32_t lvl_h[3], lvl_w[3];
    int32_t reg_stride, nc, reg_max;
    float lvl_stride[3];
    const void* aux[4];          /* DWCONV: aux[0] = Toeplitz table of the matrix-core variant (else 0)                 */
    /* engine scheduling (ignored by maf_op_launch): independent branches of the graph run on separate HIP streams */
    int32_t lane;                /* 0 = the caller's stream; 1..7 = engine-owned side streams                          */
    int32_t n_wait;              /* ops on OTHER lanes whose results this op reads (<= 8): indices into the op list     */
    int32_t wait[8];
    int32_t out_pairs;           /* CONV1X1 (fp16, tile_k = 5): store the output as pixel pairs (MAF_SRC_PAIRS) for a depth-wise consumer */
    int32_t reserved0;
} maf_op_t;

const char* maf_last_error(void);
int maf_version(void);
/* sizeof(maf_op_t) as this library was compiled: a binding checks its own struct against it at load time (a stub that is short of
 * fields would make maf_engine_create read past every element of the op array). */
int maf_op_size(void);
/* Bytes of one 32-mid-channel block record of MAF_OP_BOTTLENECK for kernel size k and c = Cin = Cout channels. */
int64_t maf_bottleneck_record_bytes(int32_t k, int32_t Cin, int32_t Cout);
/* MAF_OP_BOTTLENECK with nc = C3 > 0 (the closing 1x1 of the RepHDW block behind it): bytes of aux[0]'s record for c channels per concat slot and nsrc
 * slots read from memory (0: C3 no multiple of 16 / nsrc outside 2..3), and whether the (k, c, nsrc, C3) instantiation exists. */
int64_t maf_bottleneck_tail_record_bytes(int32_t c, int32_t nsrc, int32_t C3);
int maf_bottleneck_tail_supported(int32_t k, int32_t c, int32_t nsrc, int32_t C3);
int64_t maf_conv1dw_record_bytes(int32_t k, int32_t Cin);
int64_t maf_head_tail_record_bytes(int32_t C);
int64_t maf_stem2_record_bytes(int32_t C0, int32_t C1, int32_t C3);
int64_t maf_conv3s2_lds_record_bytes(int32_t Cin, int32_t Cout);
/* ... with the pooled 1x1 branch of MPRep behind it (MAF_OP_CONV3X3S2, tile_k = 6, nc = C1; maf-yolo_amd/pack.py:pack_mprep_lds). */
int64_t maf_mprep_lds_record_bytes(int32_t Cin, int32_t Cout, int32_t C1);
/* ... and on the tile_k = 7 kernel ((96, 96, 96); 0 = no such kernel; pack.py:pack_mprep_wreg). */
int64_t maf_mprep_wreg_record_bytes(int32_t Cin, int32_t Cout, int32_t C1);
int64_t maf_conv3s2_wreg_record_bytes(int32_t Cin, int32_t Cout);   /* 0: no instantiation for this shape */
/* MAF_OP_CONV1X1 with tile_k = 5, tile_p = 1, ONE direct source, act = NONE and aux[0] = NULL takes aux[2] = half of the scratch of the training-mode BatchNorm
 * behind the conv ([reserved0 replicas][2][Cout] fp32, the layout maf_dw_branches_stats fills; reserved0 = maf_bn_replicas(Cout, R)): the conv's epilogue adds
 * the sum and the sum of squares of the outputs it stores (after their rounding to the activation dtype), and the BatchNorm call runs as
 * maf_bn_forward_ex(..., stats_ready = 1) — no statistics pass (Conv.forward in training mode, yolov6/layers/common.py:44-47: conv -> bn -> act;
 * csrc/conv_stream_lds_st.hip).  Whether the (K / 32, tile_c) instantiation exists: */
int maf_conv1x1_stats_supported(int32_t ksteps, int32_t tile_c);

/* Launch one op on `stream`. */
int maf_op_launch(const maf_op_t* op, maf_stream_t stream);

/* A plan = ordered op list (Model.forward's node loop, yolo.py:186-201, flattened to launches).  Ops carry a lane: the
 * engine launches lane-0 ops on the caller's stream and the others on its own streams, with event waits for the cross-lane
 * reads listed in `wait` (the MAFPN neck and the three heads are independent branches: small-map layers overlap big ones);
 * all lanes fork from and join the caller's stream inside every run call, so the call keeps single-stream semantics. */
typedef struct maf_engine maf_engine_t;
int maf_engine_create(const maf_op_t* ops, int32_t n_ops, maf_engine_t** out);
int maf_engine_num_ops(const maf_engine_t* e);
/* image / pred override the STEM input and DECODE output pointers when non-NULL. */
int maf_engine_run(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream);
/* maf_engine_run + the candidate filter of non_max_suppression inside the forward: the head-tail kernels (MAF_OP_HEADTAIL) append every
 * (box, class) whose score exceeds `conf_thres` to the candidate lists of `nms_workspace` (a buffer of maf_nms_workspace_bytes(B, A, 80) bytes; its
 * counters are reset here), so that a following maf_nms_ex(pred, ..., conf_thres, ..., multi_label = 1, workspace = nms_workspace, flags |
 * MAF_NMS_PRECOLLECTED) skips its own pass over the prediction (yolov6/utils/nms.py:48,69,75-77 folded into yolov6/models/yolo.py:355-396).
 * Needs every level's tail to be a MAF_OP_HEADTAIL (MAF_E_UNSUPPORTED otherwise), H * W of every level a multiple of 16, 0 <= conf_thres < 1. */
int maf_engine_run_filtered(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream, void* nms_workspace, double conf_thres);
/* Same launches replayed from a hipGraph captured on first use (bs=1 latency path). Pointers are
 * frozen at capture time: image/pred must be the same on every call. */
int maf_engine_run_graph(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream);
/* One forward with a HIP event pair around every op; ms_per_op (HOST float[n_ops]) receives each
 * op's duration.  Synchronises the stream.  Used by bench.py for the per-kernel roofline. */
int maf_engine_run_timed(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream, float* ms_per_op);
void maf_engine_destroy(maf_engine_t* e);

/*
 * non_max_suppression (yolov6/utils/nms.py:31-105) for a whole batch in two launches.
 *   pred      fp32 [B,N,5+nc]
 *   classes   optional device int32[n_classes] filter (nms.py:83-84), NULL = none
 *   workspace device scratch of maf_nms_workspace_bytes(B,N,nc) bytes
 *   out_rows  fp32 [B,max_det,6] (x1,y1,x2,y2,conf,cls) in descending conf order
 *   out_idx   int64 [B,max_det]  flat candidate index: box*nc+cls (multi_label) or box
 *   out_count int32 [B]
 * Rules kept from the reference: strict '>' on conf_thres in fp32 (nms.py:48,76,80); candidate order
 * = row-major (box,class) (nms.py:76); conf = cls*obj (nms.py:69); boxes offset by cls*4096 unless
 * agnostic (nms.py:94-95); > 30000 candidates -> top 30000 by score (nms.py:90-91); keep <= max_det
 * (nms.py:97-98); greedy rule of torchvision.ops.nms (suppress when IoU > iou_thres, ties broken by
 * lower candidate index, the fp32 IoU compared with iou_thres in double as torchvision's CPU kernel
 * does).  conf_thres is applied in fp32 (what `tensor > python_float` does).  The 10 s wall-clock
 * break (nms.py:101-103) is dropped.  max_det <= 1024.
 */
enum { MAF_NMS_FLOAT_THRESHOLD = 1, MAF_NMS_SINGLE_LAUNCH = 2, MAF_NMS_PRECOLLECTED = 4, MAF_NMS_MATRIX = 8 };
enum { MAF_NMS_CNT_STRIDE = 64 };       /* ints between the candidate counters of two images in the NMS workspace (one 256-byte line each) */
/* maf_nms_ex flags.
 * MAF_NMS_FLOAT_THRESHOLD: compare the fp32 IoU with fl32(iou_thres) — torchvision's CUDA kernel (`float iou_threshold`), which is what
 *   yolov6/utils/nms.py:96 reaches when the reference runs on a GPU — instead of with the double (its CPU kernel, the default here and the rule of
 *   the oracle and of the golden fixtures).  The results differ only for a pair whose fp32 IoU equals fl32(iou_thres) exactly where
 *   fl32(iou_thres) > iou_thres (0.6, 0.7, ...).
 * MAF_NMS_SINGLE_LAUNCH: the whole call is ONE kernel (csrc/nms.hip:nms_single_kernel: every workgroup collects candidates, the last one to
 *   finish an image sorts and selects them) instead of seven; same results bit for bit.  Measured on configs[4] (m, bs 1): 0.29 ms against
 *   0.20 ms for the seven launches (whose pair-matrix kernel uses the whole chip) — available, not the default.
 * MAF_NMS_MATRIX: images on the all-pairs path (<= 4096 candidates, no per-class split) build the n x n suppression bit matrix on the whole chip and scan its rows
 *   (rounds 2-5's form, two launches) instead of the kept-list scan (one workgroup per image tests every block of 64 sorted candidates against the boxes kept so
 *   far only; the default since round 6).  Same survivors bit for bit; kept for A/B.
 * MAF_NMS_PRECOLLECTED: the candidate lists of `workspace` (counters + keys) are already there — written by the forward pass that produced `pred`
 *   (maf_engine_run_filtered: the head-tail kernels test the scores they have just computed against the same conf_thres) — so the call skips its
 *   counter reset and its pass over the prediction tensor.  Only for multi_label with nc > 1, no class filter, conf_thres < 1. */
int64_t maf_nms_workspace_bytes(int32_t B, int32_t N, int32_t nc);
int maf_nms(const float* pred, int32_t B, int32_t N, int32_t nc, double conf_thres, double iou_thres,
            const int32_t* classes, int32_t n_classes, int32_t agnostic, int32_t multi_label,
            int32_t max_det, void* workspace, int64_t workspace_bytes,
            float* out_rows, int64_t* out_idx, int32_t* out_count, maf_stream_t stream);
int maf_nms_ex(const float* pred, int32_t B, int32_t N, int32_t nc, double conf_thres, double iou_thres,
               const int32_t* classes, int32_t n_classes, int32_t agnostic, int32_t multi_label,
               int32_t max_det, void* workspace, int64_t workspace_bytes,
               float* out_rows, int64_t* out_idx, int32_t* out_count, int32_t flags, maf_stream_t stream);

/*
 * Training-side entry points.  The train-form graph keeps conv and BatchNorm apart
 * (yolov6/layers/common.py:46-47, 219-224, 3024-3031), so the convs run through maf_op_launch with a zero bias and
 * act = none; their backward needs:
 *   maf_pack_w1x1   fp32 weight [Cout][Cin] of nn.Conv2d(k=1) -> the fragment-packed operand of MAF_OP_CONV1X1 in `dtype`;
 *                   transpose = 1 packs W^T, which turns the same kernel into the data gradient dX = dY * W.
 *   maf_pack_dw     fp32 depth-wise weight [C][k*k] -> [k*k][C] in `dtype`; flip = 1 gives the data-gradient kernel.
 *   maf_dw_wgrad    dW[c][ky][kx] += sum over pixels of dY * shifted X.  dw = `replicas` zeroed copies [replicas][C][k*k] fp32: the
 *                   workgroups spread their atomics over the copies (atomics on one cache line serialise), the caller sums them.
 *   maf_conv1x1_wgrad  dW[co][ci] += sum over the M = B*H*W pixels of dY[m][co] * X[m][ci]  (fp16 NHWC views, fp32 result
 *                   accumulated with atomics: dw must be zeroed by the caller) — pixel chunks per workgroup, tiles transposed
 *                   through LDS, MFMA; replaces the TN GEMM with a tiny output and a huge reduction.
 */
int64_t maf_pack_w1x1_bytes(int32_t Cout, int32_t Cin, int32_t transpose, int32_t dtype, int32_t tile_c);
int maf_pack_w1x1(const float* w, int32_t Cout, int32_t Cin, int32_t transpose, int32_t dtype, int32_t tile_c, void* out, maf_stream_t stream);
int maf_pack_dw(const float* w, int32_t C, int32_t k, int32_t flip, int32_t dtype, void* out, maf_stream_t stream);
/* Task-aligned label assignment on RAGGED targets, no host round trip.
 *   maf_tal_targets  replaces ComputeLoss.preprocess (yolov6/models/loss.py:179-188: python lists + targets.cpu().numpy()):
 *                    targets [T][6] = (image, class, cx, cy, w, h) normalised -> gts [T][5] = (label, x1, y1, x2, y2 in pixels) grouped by
 *                    image in their original order, gt_image [T], offsets [B+1] = first row of every image.  Rows whose image id is
 *                    outside [0, B) are dropped (offsets[B] rows are valid).  T <= 16384, B <= 4096.
 *   maf_tal_assign   replaces TaskAlignedAssigner.forward (yolov6/assigners/tal_assigner.py:21-151, assigner_utils.py:25-89) as
 *                    ComputeLoss calls it (loss.py:96-103).  pd_scores [B,A,nc] (sigmoid outputs, score_dtype MAF_F16 / MAF_F32),
 *                    pd_bboxes [B,A,4] xyxy pixels fp32, cand_scratch: T*topk int32.  Anchors: n_levels <= 4 grids, level_hw [n_levels][2]
 *                    (host ints, rows x columns), level_stride [n_levels] (host floats), row-major per level, levels concatenated;
 *                    anchor_points [A,2] (device) = (cell + cell_offset) * stride as generate_anchors builds them.  out_gt [B,A] = row of
 *                    the assigned box or -1 (background), out_norm [B,A] = the normalised alignment metric that scales the one-hot
 *                    score target (tal_assigner.py:66-71).  A <= 8400; at most 8400 boxes per image are considered. */
int maf_tal_targets(const float* targets, int32_t T, int32_t B, float img_size, float* gts, int32_t* gt_image, int32_t* offsets, maf_stream_t stream);
int maf_tal_assign(const void* pd_scores, int32_t score_dtype, const float* pd_bboxes, const float* anchor_points, const float* gts,
                   const int32_t* gt_image, const int32_t* offsets, int32_t T, int32_t B, int32_t A, int32_t nc, int32_t topk,
                   float alpha, float beta, float eps, int32_t n_levels, const int32_t* level_hw, const float* level_stride, float cell_offset,
                   int32_t* cand_scratch, int32_t* out_gt, float* out_norm, maf_stream_t stream);
/* The warm-up assigner of ComputeLoss (epoch < warmup_epoch, loss.py:83-91) — replaces ATSSAssigner.forward (yolov6/assigners/atss_assigner.py:17-161,
 * iou2d_calculator.py:63-246, assigner_utils.py:4-66) on the same ragged targets and with the same outputs as maf_tal_assign:
 * out_norm = IoU of the predicted box with the assigned box (the soft label, atss_assigner.py:80-84).  topk = 9 nearest anchors per
 * level, cell_size = 5 (side of the square anchor boxes in strides, anchor_generator.py:29), cand_scratch: T * n_levels * topk int32;
 * every level needs at least 3 x 3 anchors (the reference raises below 9 anchors on a level); box centres are expected inside the image. */
int maf_atss_assign(const float* pd_bboxes, const float* anchor_points, const float* gts, const int32_t* gt_image, const int32_t* offsets,
                    int32_t T, int32_t B, int32_t A, int32_t topk, int32_t n_levels, const int32_t* level_hw, const float* level_stride,
                    float cell_offset, float cell_size, int32_t* cand_scratch, int32_t* out_gt, float* out_norm, maf_stream_t stream);
/* The loss terms of ComputeLoss (yolov6/models/loss.py:56-267, task-aligned branch) from the assigner's two arrays; head outputs in
 * `dtype` (MAF_F16 under autocast, or MAF_F32); reg_max must be 16; anchor_strides [A] = stride of every anchor.
 *   maf_loss_decode  bbox_decode (:190-193) * stride: pred_distri [B,A,4*17] logits -> out_boxes [B,A,4] xyxy pixels fp32.
 *   maf_loss_terms   VarifocalLoss (:196-206) over all scores + GIoU (figure_iou.py) and DFL (:209-267) over the foreground anchors.
 *     forward form   (upstream, grad_* NULL): out[5] (device) = weighted total, w_iou*iou, w_dfl*dfl, w_cls*cls (the order of the
 *                    reference's loss items), target-score sum; partials = scratch of maf_loss_partial_rows(B,A,nc) float4 rows.
 *     gradient form  upstream = device scalar d(objective)/d(total); out = what the forward form wrote; grad_scores / grad_distri
 *                    (shape and dtype of the inputs) receive d(objective)/d(input). */
int64_t maf_loss_partial_rows(int32_t B, int32_t A, int32_t nc);
int maf_loss_decode(const void* pred_distri, int32_t dtype, const float* anchor_points, const float* anchor_strides, int32_t B, int32_t A,
                    int32_t reg_max, float* out_boxes, maf_stream_t stream);
int maf_loss_terms(const void* pred_scores, const void* pred_distri, int32_t dtype, const float* anchor_points, const float* anchor_strides,
                   const float* gts, const int32_t* assigned_gt, const float* norm, int32_t B, int32_t A, int32_t nc, int32_t reg_max,
                   float w_cls, float w_iou, float w_dfl, float* partials, float* out, const float* upstream, void* grad_scores,
                   void* grad_distri, maf_stream_t stream);

/* BatchNorm2d in training mode fused with the activation behind it (Conv.forward = act(bn(conv(x))), common.py:46-47), NHWC views.
 *   maf_bn_forward   batch statistics -> save_mean / save_rstd (+ running stats with torch's momentum rule, unbiased variance, and
 *                    the int64 num_batches_tracked counter += 1; each may be NULL) and y = act(xhat*gamma + beta); two launches (statistics, apply).
 *   maf_bn_backward  dz = gradient w.r.t. the activation output; recomputes u from x; dx, dgamma, dbeta; two launches.
 *   residual         (may be NULL) a tensor like y added BEFORE the activation: y = act(xhat*gamma + beta + residual) — the branch sums of
 *                    RepVGGBlock (common.py:224) and DilatedReparamBlock (:3028-3031) without their own pass.  Backward needs it only when the
 *                    activation does (ReLU): then dres receives dz * act'(u); without an activation the residual's gradient IS dz (pass NULL).
 *   part             fp32 scratch of TWO halves [2][R][2][roundup(C,256)], zeroed ONCE by the caller: a call accumulates its partial sums
 *                    into half `phase` (R replicas spread the atomics) and clears half 1 - phase — the one the previous call on the stream
 *                    used — so one buffer per stream serves all BatchNorms of a step when the caller alternates phase = 0, 1, 0, ... */
int maf_bn_forward(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t act, void* y,
                   int32_t y_stride, float* save_mean, float* save_rstd, float* part, int32_t R, int32_t phase, const void* residual, int32_t res_stride,
                   maf_stream_t stream);
int maf_bn_backward(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, int32_t M, int32_t C, int32_t dtype,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int32_t act,
                    void* dx, int32_t dx_stride, float* dgamma, float* dbeta, float* part, int32_t R, int32_t phase,
                    const void* residual, int32_t res_stride, void* dres, int32_t dres_stride, maf_stream_t stream);
/* The parallel depth-wise branches of a train-form DilatedReparamBlock (yolov6/layers/common.py:3024-3031: lk_origin(x) and dil_conv_k*(x), NB
 * depth-wise "same" convolutions of ONE input with kernel sizes 3, 3, 1 / 5, 3, 1 / 7, 5, 3 / 9, 7, 5, 3 for k0 = 3 / 5 / 7 / 9 — the 1 x 1 branch is a
 * per-channel scale) in one launch per direction, NHWC, no bias, no activation (a BatchNorm follows each):
 *   dgrad = 0   dst[j] = DW(src[0], w[j]), j < nb            (the input is staged once)
 *   dgrad = 1   dst[0] = sum_j DW(src[j], w[j])             (w[j] = the flipped filters; the sum is kept in fp32 registers and written once)
 * (k0, nb) in {(3, 3), (5, 3), (7, 3), (9, 4)}; w[j]: [k_j * k_j][C] in `dtype` (maf_pack_dw); strides in elements, multiples of the 16-byte
 * channel group; src / dst arrays hold 1 / nb (forward) or nb / 1 (data gradient) entries. */
/* Forward of maf_dw_branches + the BatchNorm statistics of every branch: stats[j] (may be NULL per branch) = the half of the `part` scratch that the
 * maf_bn_forward_ex(..., stats_ready = 1) call of branch j's BatchNorm will read, [replicas][2][C] fp32, zero on entry; the kernel adds the sum and
 * the sum of squares of the (rounded) outputs it stores, spread over `replicas` = maf_bn_replicas(C, R) copies. */
int maf_dw_branches_stats(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                          int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, float* const* stats, int32_t replicas, maf_stream_t stream);
int maf_dw_branches(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                    int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t dgrad, maf_stream_t stream);
/* maf_bn_forward with stats_ready != 0: half `phase` of `part` already holds this call's {sum x, sum x^2} — accumulated by the kernel that produced x
 * (maf_dw_branches_stats) into replicas [0, maf_bn_replicas(C, R)) — and the statistics launch is skipped: apply only. */
int maf_bn_forward_ex(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t act, void* y,
                      int32_t y_stride, float* save_mean, float* save_rstd, float* part, int32_t R, int32_t phase, const void* residual, int32_t res_stride,
                      int32_t stats_ready, maf_stream_t stream);
int32_t maf_bn_replicas(int32_t C, int32_t R);
/* The statistics pass of maf_bn_forward alone (half `phase` of `part` += {sum x, sum x^2}; nothing is cleared). */
int maf_bn_stats(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, float* part, int32_t R, int32_t phase, maf_stream_t stream);
/* The SUM of the nb (2..4) training-mode BatchNorm2d of a DilatedReparamBlock (common.py:3024-3031: origin_bn(lk_origin(x)) + sum_j dil_bn_j(dil_conv_j(x)),
 * no activation; act = MAF_ACT_RELU: RepVGGBlock's ReLU(BN(3x3 s2) + BN(1x1 s2)), common.py:224 — the backward recomputes the sum for the ReLU's mask)
 * as one apply pass per direction (csrc/bn_sum.hip):
 *   maf_bn_sum_forward   out = sum_j BN_j(z_j).  part[j] = branch j's own scratch ([2][R][2][roundup(C,256)], the layout of maf_bn_forward), whose half
 *                        phase[j] ALREADY holds {sum z_j, sum z_j^2} (maf_dw_branches_stats / maf_bn_stats); save_mean / save_rstd / running statistics /
 *                        num_batches_tracked per branch as in maf_bn_forward; the other half of every scratch is cleared.
 *   maf_bn_sum_backward  dy = gradient of the sum (the same for every branch): one statistics launch ({sum dy, sum dy xhat_j}), one apply launch writing
 *                        every dz_j; dgamma_j / dbeta_j written, or added when accumulate_affine != 0.  bpart = [2][R][1 + nb][roundup(C,256)] fp32, zeroed
 *                        once by the caller, halves alternating call by call (phase) like maf_bn_backward's scratch. */
int maf_bn_sum_forward(const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                       const float* const* gamma, const float* const* beta, float eps, float momentum,
                       float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                       void* out, int32_t out_stride, float* const* save_mean, float* const* save_rstd,
                       float* const* part, int32_t R, const int32_t* phase, int32_t act, maf_stream_t stream);
/* maf_bn_sum_forward that ALSO accumulates the statistics of the values it stores for the training-mode BatchNorm that normalises `out` next (UniRepLKNetBlock.norm behind
 * the DilatedReparamBlock, yolov6/layers/common.py:3053-3083): half next_phase of next_part ([2][next_R][2][roundup(C,256)], that BatchNorm's own scratch) += {sum out,
 * sum out^2}; its call is then the apply pass alone (maf_bn_forward_ex, statistics ready).  next_part = NULL: maf_bn_sum_forward. */
int maf_bn_sum_forward_stats(const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                             const float* const* gamma, const float* const* beta, float eps, float momentum,
                             float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                             void* out, int32_t out_stride, float* const* save_mean, float* const* save_rstd,
                             float* const* part, int32_t R, const int32_t* phase, int32_t act,
                             float* next_part, int32_t next_R, int32_t next_phase, maf_stream_t stream);
int maf_bn_sum_backward(const void* dy, int32_t dy_stride, const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                        const float* const* gamma, const float* const* beta, const float* const* save_mean, const float* const* save_rstd,
                        void* const* dz, const int32_t* dz_stride, float* const* dgamma, float* const* dbeta, int32_t accumulate_affine,
                        float* bpart, int32_t R, int32_t phase, int32_t act, maf_stream_t stream);
/* maf_bn_backward with accumulate_affine != 0: dgamma / dbeta are ADDED to what the buffers hold — the slices of a gradient-exchange bucket
 * (maf_yolo_amd/exchange.py: `p.grad` of the BatchNorm affine parameters is a view of a flat fp32 bucket; the reference accumulates them with
 * AccumulateGrad, i.e. 280 one-line add kernels per step of MAF-YOLO-n, yolov6/core/engine.py:164). */
int maf_bn_backward_acc(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, int32_t M, int32_t C, int32_t dtype,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int32_t act,
                        void* dx, int32_t dx_stride, float* dgamma, float* dbeta, float* part, int32_t R, int32_t phase,
                        const void* residual, int32_t res_stride, void* dres, int32_t dres_stride, int32_t accumulate_affine, maf_stream_t stream);
/* Bit-reproducible BatchNorm statistics (a test / debugging mode, process-wide, one stream at a time): the statistics kernels of maf_bn_forward /
 * maf_bn_backward write per-workgroup sums with plain stores and a third launch adds them in a fixed order, instead of fp32 atomics whose order
 * changes from run to run.  With it two forward passes of the train-form graph are bit-identical (torch.use_deterministic_algorithms' role). */
int maf_set_deterministic(int32_t on);
int maf_conv1x1_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t M, int32_t Cin, int32_t Cout,
                      int32_t dtype, float* dw, maf_stream_t stream);
/* General form: dW (fp32, accumulated into) of a conv with k = 1 (stride 1 / 2, pad 0) or k = 3 (stride 2, pad 1) — RepVGGBlock.rbr_dense /
 * rbr_1x1 (common.py:202-203), ConvWrapper (:76-83): x [B,Hs,Ws,Cin], dy [B,Ho,Wo,Cout], fp16 NHWC views.  The taps are gathered inside the
 * kernel (no im2col tensor); inputs wider than 256 channels run as channel chunks.
 * Layout of dW: k = 1 -> [Cout][Cin];  k = 3 -> TAP-MAJOR [3][3][Cout][Cin] (the lanes of an atomic then share a cache line). */
int maf_conv_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t B, int32_t Ho, int32_t Wo, int32_t Hs, int32_t Ws,
                   int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t dtype, float* dw, maf_stream_t stream);
int maf_dw_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t B, int32_t H, int32_t W, int32_t C,
                 int32_t k, int32_t dtype, float* dw, int32_t replicas, maf_stream_t stream);
/* Training-form stem (backbone.0's RepVGGBlock over the image, yolov6/layers/common.py:199-203, 219-224; Cin = 3: configs/yaml/MAF-YOLO-n.yaml:5): z3 = conv3x3 stride 2 pad 1
 * (img, w3) and z1 = conv1x1 stride 2 (img, w1) in ONE launch, no bias, no activation (csrc/stem_train.hip).  img: fp16 NHWC [B][Hin][Win] with 16-byte pixels (3 channels
 * zero-padded to 8; pixel stride img_stride halfs), w3 [Cout][3][3][3] / w1 [Cout][3] fp32 — the parameters themselves, rounded to fp16 inside as autocast does —, z3 / z1
 * dense fp16 [B][Hin/2][Win/2][Cout]. */
int maf_stem_train(const void* img, int32_t img_stride, int32_t B, int32_t Hin, int32_t Win, const float* w3, const float* w1, int32_t Cout,
                   void* z3, void* z1, maf_stream_t stream);
/* Input staging of the training step: contiguous NCHW images [B][3][H][W] (MAF_F32 or MAF_F16; the reference's `images.float() / 255`, yolov6/core/engine.py:426) -> the
 * NHWC fp16 buffer with 16-byte pixels that maf_stem_train and the stem's weight-gradient kernels read: out [B][H][W][8], channels 3..7 zero.  One pass. */
int maf_image_to_nhwc8(const void* x, int32_t dtype, int32_t B, int32_t H, int32_t W, void* out, maf_stream_t stream);
/* The 3x3 (+ second 3x3) + 1x1 branches of a train-form DilatedReparamBlock (kernel sets 3,3,1 / 5,3,1: yolov6/layers/common.py:2997-3008, 3024-3031) share their
 * input: ONE launch stages the X halo tile once and multiplies it with dYa (and dYb, may be null together with dwb) for the 3x3 gradients [C][9] and — the items of
 * the centre tap row — with dY1 for the 1x1 branch's per-channel scale gradient dw1[C].  Same contract as maf_dw_wgrad otherwise (zeroed fp32 results, `replicas` copies). */
int maf_dw_wgrad31(const void* x, int32_t x_stride, const void* dya, int32_t dya_stride, const void* dyb, int32_t dyb_stride, const void* dy1, int32_t dy1_stride,
                   int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, float* dwa, float* dwb, float* dw1, int32_t replicas, maf_stream_t stream);

/* Gradient fold of the owned gradient exchange (maf_yolo_amd/exchange.py; the reference leaves this to autograd's AccumulateGrad + the DDP
 * reducer's bucket copy, yolov6/core/engine.py:161-164): src = what a weight-gradient kernel produced, tap-major [taps][Cout_p][Cin_p] fp32 with
 * padded channel counts; dst = the parameter-shaped slice [Cout][Cin][taps] of a flat gradient bucket; accumulate != 0 adds (gradient
 * accumulation over several backward passes), 0 overwrites.  Runs on `stream` (the weight-gradient stream). */
int maf_grad_fold(const float* src, int32_t taps, int32_t Cout_p, int32_t Cin_p, float* dst, int32_t Cout, int32_t Cin, int32_t accumulate,
                  maf_stream_t stream);
/* dst[b, 2y, 2x, :] += src[b, y, x, :]: src [B,Ho,Wo,C], dst [B,2Ho,2Wo,C] NHWC with pixel strides in elements — the data gradient of RepVGGBlock's stride-2 1x1
 * branch (yolov6/layers/common.py:203, 224) added onto the 3x3 branch's data gradient in place. */
int maf_add_sub2(const void* src, int32_t src_stride, void* dst, int32_t dst_stride, int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, maf_stream_t stream);
/* out[c] += sum over the M pixels of x[m * x_stride + c] (fp32 atomics; the caller zeroes or accumulates): the bias gradient of the head's prediction convs
 * (nn.Conv2d with bias, yolov6/layers/common.py:1304-1305).  C <= 256. */
int maf_colsum(const void* x, int32_t x_stride, int64_t M, int32_t C, int32_t dtype, float* out, maf_stream_t stream);

/* Zero `bytes` bytes at p on `stream` (hipMemsetAsync): the fp32 accumulation buffers of maf_conv_wgrad / maf_dw_wgrad when those run on a stream
 * of their own. */
int maf_zero(void* p, int64_t bytes, maf_stream_t stream);

/* MaxPool2d(k, stride, padding) of the training graph — SPPF.m (5, 1, 2; yolov6/layers/common.py:114-129) and MP (2, 2, 0; :667-673) —
 * NHWC views, fp16 / fp32, floor mode: x [B,H,W,C] -> y [B,Ho,Wo,C], Ho = floor((H + 2 pad - k) / stride) + 1.  Forward keeps the chosen
 * window element (row * k + column, first maximum in scan order like aten) as one byte per element in idx [B,Ho,Wo,C]; backward is a
 * gather over the windows that contain an input element (no atomics). */
int maf_maxpool_forward(const void* x, int32_t x_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                        int32_t dtype, void* y, int32_t y_stride, uint8_t* idx, maf_stream_t stream);
int maf_maxpool_backward(const void* dy, int32_t dy_stride, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                         int32_t stride, int32_t pad, int32_t dtype, void* dx, int32_t dx_stride, maf_stream_t stream);
/* nn.Upsample(scale_factor=2, mode="nearest") of the training graph (the neck's two up-sampling nodes, configs/yaml/MAF-YOLO-n.yaml) on NHWC views with pixel strides
 * in elements — source and destination may be channel slices (slots of a concat buffer): y [B,2H,2W,C] from x [B,H,W,C]; backward dx = the sum of the four. */
int maf_upsample2x_forward(const void* x, int32_t x_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* y, int32_t y_stride, maf_stream_t stream);
int maf_upsample2x_backward(const void* dy, int32_t dy_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* dx, int32_t dx_stride, maf_stream_t stream);

/* Weight staging of a whole training step in one launch.  A descriptor transforms one fp32 weight tensor:
 *   kind 0  dense conv weight [Cout][Cin][taps] (taps = 1 or 9) -> the MFMA fragment order of maf_pack_w1x1 with tile_c = CT; K runs
 *           tap-major, every tap padded to Kp (a multiple of the k-step: 32 fp16 / 16 fp32), steps = taps * Kp / k-step for 3x3 and
 *           ceil(K / k-step) for 1x1 (then Kp = steps * k-step); transpose = 1 packs W^T (the data gradient's operand)
 *   kind 1  depth-wise kernel [C = Cout][taps = k*k] -> [k*k][C], flip = 1 reverses the taps (data gradient)
 *   kind 2  fp32 vector of Cout values -> `total` fp32 values, zero behind the Cout-th (the bias of a prediction conv padded to the conv's channel tile)
 * total = packed elements, block0 = first block of the descriptor in the flattened grid (1024 elements per block; ascending),
 * nblocks = sum of ceil(total / 1024).  The descriptor array lives in DEVICE memory. */
typedef struct maf_pack_desc {
    const void* src; void* dst;
    int64_t total;
    int32_t kind, dtype;
    int32_t Cout, Cin, taps, transpose;
    int32_t CT, steps, Kp, flip;
    int32_t block0, reserved;
} maf_pack_desc_t;
int maf_pack_batch(const maf_pack_desc_t* descs_dev, int32_t n, int32_t nblocks, maf_stream_t stream);
int32_t maf_pack_desc_size(void);

/* ModelEMA.update (yolov6/utils/ema.py:25-37: `v *= d; v += (1 - d) * msd[k].detach()` for every floating-point state_dict entry) in one
 * launch over a descriptor table in DEVICE memory: dst (the average) and src (the model) fp32, `total` elements, block0 = first block of the
 * entry in the flattened grid (1024 elements per block; ascending), nblocks = sum of ceil(total / 1024).  decay and one_minus_decay are the
 * two scalars as the reference's tensor ops see them (Python doubles rounded to fp32); the two products and the sum are rounded
 * separately, so the result is bit-identical to the reference's three element-wise ops. */
typedef struct maf_ema_desc {
    void* dst; const void* src;
    int64_t total;
    int32_t block0, reserved;
} maf_ema_desc_t;
int maf_ema_update(const maf_ema_desc_t* descs_dev, int32_t n, int32_t nblocks, float decay, float one_minus_decay, maf_stream_t stream);
int32_t maf_ema_desc_size(void);

/* The optimizer step of the train loop (yolov6/core/engine.py:375-391: `self.scaler.step(self.optimizer)`; optimizer = torch.optim.SGD(momentum, nesterov=True) over
 * the three groups of yolov6/solver/build.py:12-33) for every parameter in ONE launch over a descriptor table in DEVICE memory: param / grad / buf fp32 (buf = the
 * momentum buffer, NULL for momentum 0), `total` elements, block0 as in the EMA table above — 1024 elements per block, group = index into the per-group hyper-parameters
 * (host arrays of ngroups <= MAF_SGD_MAX_GROUPS doubles: what the framework's optimizer holds).  found_inf / grad_scale: the GradScaler's device scalars (NULL = none) —
 * found_inf == 1 skips the update, grad_scale un-scales the gradients (and they are written back un-scaled), no host round trip.  Per element, in double where the
 * framework's fused kernel computes in double (each multiply-add one fused operation, as its build contracts them):  g = g / scale;  g += wd * p;  buf = mu * buf + g;
 * g = nesterov ? g + mu * buf : buf;  p -= lr * g. */
#define MAF_SGD_MAX_GROUPS 8
typedef struct maf_sgd_desc {
    void* param; void* grad; void* buf;
    int64_t total;
    int32_t block0, group;
} maf_sgd_desc_t;
int maf_sgd_update(const maf_sgd_desc_t* descs_dev, int32_t n, int32_t nblocks, int32_t ngroups, const double* lr, const double* weight_decay, const double* momentum,
                   const int32_t* nesterov, const float* found_inf, const float* grad_scale, maf_stream_t stream);
int32_t maf_sgd_desc_size(void);

/* The inf check GradScaler.step runs over the optimizer's gradients before the step (yolov6/core/engine.py:375-391) in one launch over a table of the contiguous
 * fp32 ranges they occupy, in DEVICE memory: ptr, `total` elements, block0 = first block of the range in the flattened grid, 4096 elements per block (ascending;
 * nblocks = sum of ceil(total / 4096)).  *found_inf (a device float the caller has zeroed) becomes 1 if any element is Inf or NaN. */
typedef struct maf_range_desc {
    const void* ptr;
    int64_t total;
    int32_t block0, reserved;
} maf_range_desc_t;
int maf_nonfinite_check(const maf_range_desc_t* descs_dev, int32_t n, int32_t nblocks, float* found_inf, maf_stream_t stream);
int32_t maf_range_desc_size(void);


/* dst = [dst +] sum_i src[i] over NHWC views with pixel strides in elements (channel slices of wider buffers are fine): n = 1, accumulate = 0 is a strided copy
 * (a concat input its producer could not store in place: torch.cat of the neck, configs/yaml/MAF-YOLO-n.yaml:16-42), n = 1, accumulate = 1 an in-place add
 * (a tensor that feeds a concat AND a later block: the block's gradient added into the concat's), n = 2..4 the gradient of a tensor with several consumers
 * (what autograd's engine does with one add kernel per extra consumer).  fp16 (fp32 sums, one rounding) / fp32; C and strides in whole 16-byte groups. */
int maf_nhwc_sum(const void* const* src, const int32_t* src_stride, int32_t n, void* dst, int32_t dst_stride, int64_t M, int32_t C, int32_t dtype,
                 int32_t accumulate, maf_stream_t stream);
/* Detect_yaml.forward, TRAIN branch (yolov6/models/yolo.py:333-354) with the head's class sigmoid (yolov6/layers/common.py:1332) folded in: per level l the NHWC maps
 * cls[l] (class LOGITS, nc channels, pixel stride cls_stride[l] elements) and reg[l] (nreg = 4 * (reg_max + 1) channels) of B images with level_pixels[l] = h*w
 * pixels each become rows [a0_l, a0_l + h*w) of cls_out [B, A, nc] = sigmoid(logits) and reg_out [B, A, nreg] (A = sum of level_pixels) — the reference's
 * flatten(2).permute(0, 2, 1) + torch.cat.  f16 / f32; nc, nreg and the strides multiples of 4.  Host arrays are read at call time.
 * Backward: d logits = d_cls * y * (1 - y) (y = cls_out: what the framework's sigmoid backward computes from its saved output) and d_reg scattered back into per-level
 * NHWC gradient maps of nc_pad / nreg_pad channels (>= nc / nreg, the 16-byte group the conv kernels read; pad channels are WRITTEN as zeros).  d_cls / d_reg NULL =
 * no gradient (zeros). */
int maf_detect_join(const void* const* cls, const int32_t* cls_stride, const void* const* reg, const int32_t* reg_stride, const int32_t* level_pixels,
                    int32_t n_levels, int32_t B, int32_t nc, int32_t nreg, int32_t dtype, void* cls_out, void* reg_out, maf_stream_t stream);
int maf_detect_join_backward(const void* d_cls, const void* d_reg, const void* cls_out, const int32_t* level_pixels, int32_t n_levels, int32_t B, int32_t nc,
                             int32_t nreg, int32_t dtype, void* const* dcls, const int32_t* dcls_stride, void* const* dreg, const int32_t* dreg_stride,
                             int32_t nc_pad, int32_t nreg_pad, maf_stream_t stream);
/* `side` waits for what `main` holds now (an event record + a stream wait: the fork of a weight gradient to its own stream); join: `main` waits for `side`. */
int maf_stream_fork(maf_stream_t main, maf_stream_t side);
int maf_stream_join(maf_stream_t main, maf_stream_t side);

/* A stream restricted to the compute units of a bit mask (n_words x 32 bits; bit i = CU i in the driver's numbering): bench.py's serving loop can confine the
 * NMS of batch i (yolov6/utils/nms.py:31-105) to a slice of the chip while the forward of batch i + 1 runs on the rest.  Destroy with maf_stream_destroy. */
int maf_stream_create_masked(const uint32_t* mask_words, int32_t n_words, maf_stream_t* out);
int maf_stream_destroy(maf_stream_t s);

/*
 * Step tape: a recorded launch list of the C-ABI calls of one training step (yolov6/core/engine.py:141-167: forward, backward), replayed by ONE call.
 * The train-form graph is ~420 launches forward and ~470 backward of kernels that take 5-60 us each; issued one by one from Python autograd
 * Functions the host needs 19-27 ms per step — the step itself.  The Python side (maf_yolo_amd/tape.py) records every call of a step — entry point,
 * arguments, stream — once per (model, batch shape) while it runs the step the normal way with every buffer kept alive, and later steps replay the list:
 * the kernels, their order and their streams are exactly the recorded ones, launched eagerly (a hipGraph node costs ~1.1 us more than an eager launch
 * on this part).
 *   fn      index into the table of tape-able entry points (maf_tape_fn_id(name)); MAF_TAPE_FORK = maf_stream_fork(streams[a[0]], streams[a[1]])
 *   stream  index into the `streams` array of maf_tape_run the entry's last argument (its maf_stream_t) is taken from: 0 the main stream, 1 the weight-gradient
 *           stream, 2.. lanes (independent branches of the graph — the two branches of a detection head — recorded on streams of their own)
 *   a[]     the arguments in declaration order, one 64-bit slot each: pointers and integers as they are, a float as its 32 bits; pointer-to-array
 *           arguments point at host arrays the recorder keeps alive
 * Toggles: words that alternate from step to step (the phase of a BatchNorm scratch: which half this step accumulates into) are XOR-ed with their
 * mask by maf_tape_toggle between two runs.
 */
#define MAF_TAPE_MAX_ARGS 28
#define MAF_TAPE_FORK (-1)
typedef struct maf_tape_rec {
    int32_t fn, stream;
    uint64_t a[MAF_TAPE_MAX_ARGS];
} maf_tape_rec_t;
typedef struct maf_tape_toggle {
    void* addr; uint64_t mask; int32_t width, reserved;     /* width 4 or 8 bytes */
} maf_tape_toggle_t;
int32_t maf_tape_fn_id(const char* name);                   /* -1000: not a tape-able entry point */
int32_t maf_tape_fn_nargs(int32_t fn);
int32_t maf_tape_rec_size(void);
int maf_tape_run(const maf_tape_rec_t* recs, int32_t first, int32_t last, const maf_stream_t* streams, int32_t n_streams, int32_t* failed_at);
int maf_tape_toggle(const maf_tape_toggle_t* t, int32_t n);

/*
 * Post-NMS tail — replaces Evaler.scale_coords (yolov6/core/evaler.py:382-409, ratio_pad branch), box_convert
 * (:374-381) and the tensor part of convert_to_coco_format (:411-420) for a whole batch in one launch.
 *   rows [B,max_det,6] / count [B]   the outputs of maf_nms
 *   img_params [B][6] fp32           original h0, w0, gain applied to x, gain applied to y, pad_w, pad_h  (the reference divides
 *                                    x by gain[1] only with scale_exact, else both axes by gain[0])
 *   ids [n_ids]                      class index -> dataset category id (COCO 80 -> 91 table), may be NULL with n_ids = 0
 *   out [sum(count)][7] fp32         image index in the batch, category id, x, y, w, h (original-image pixels), score;
 *                                    rows of image b follow those of image b-1; *out_total = number of rows
 */
int maf_coco_rows(const float* rows, const int32_t* count, int32_t B, int32_t max_det, const float* img_params,
                  const int32_t* ids, int32_t n_ids, float* out, int32_t* out_total, maf_stream_t stream);

/* Diagnostics: shader-clock cycle stamps of image 0 of the last maf_nms call (synchronises the device):
 * [0] sort, [1] kept-list screening, [2] wave resolution, [3] total, [4] candidates, [5] survivors. */
int maf_nms_debug(uint64_t* host8);

/* HIP-event timing helper used by bench.py (events recorded on `stream`, not torch's). */
int maf_timer_create(void** timer);
int maf_timer_start(void* timer, maf_stream_t stream);
int maf_timer_stop(void* timer, maf_stream_t stream);
int maf_timer_elapsed_ms(void* timer, float* ms);   /* synchronises on the stop event */
void maf_timer_destroy(void* timer);

#ifdef __cplusplus
}
#endif
#endif
