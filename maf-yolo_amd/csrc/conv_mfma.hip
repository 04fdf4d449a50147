// Argument checking + dispatch for the MFMA conv kernels (kernels: conv_mfma.inc.h)
#include <cstdlib>
#include "conv_mfma.inc.h"

int maf_launch_conv_mfma(const maf_op_t* op, hipStream_t s) {
    if (op->kind == MAF_OP_CONV3X3S2 && op->nc && op->tile_k != 6 && op->tile_k != 7) { maf_set_error("conv3x3s2: the pooled 1x1 branch (nc) exists in the tile_k = 6 and 7 kernels only"); return MAF_E_UNSUPPORTED; }
    if (op->kind == MAF_OP_CONV3X3S2 && op->tile_k == 6) return maf_launch_conv3s2_lds(op, s);   // narrow layers: weights + input patch in LDS (conv3s2_lds.hip)
    if (op->kind == MAF_OP_CONV3X3S2 && op->tile_k == 7) return maf_launch_conv3s2_wreg(op, s);  // 96 / 128 channels: weights in registers, patch by DMA (conv3s2_wreg.hip)
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "conv: dtype must be f16/f32");
    const int CH = op->dtype == MAF_F16 ? 8 : 4, KS = op->dtype == MAF_F16 ? 32 : 16;
    MAF_REQUIRE(op->nsrc >= 1 && op->nsrc <= 4, "conv: nsrc must be 1..4");
    MAF_REQUIRE(op->B > 0 && op->H > 0 && op->W > 0 && op->Cout > 0, "conv: bad dims");
    MAF_REQUIRE(op->Cout % 2 == 0, "conv: Cout must be even");
    MAF_REQUIRE(op->out_stride % 4 == 0 && op->out_coff % 4 == 0, "conv: out stride/coff must be multiples of 4");
    MAF_REQUIRE(op->tile_c != 8 || op->dtype != MAF_F16 || op->out_f32 || (op->out_stride % 8 == 0 && op->out_coff % 8 == 0), "conv: 16-byte stores need out stride/coff multiples of 8");
    MAF_REQUIRE((long long)op->B * op->H * op->W < (1ll << 31), "conv: too many pixels");
    ConvArgs a = {};
    int csum = 0, ksum = 0;
    a.cum[0] = 0;
    bool any_special = false;
    for (int i = 0; i < 4; ++i) {
        if (i < op->nsrc) {
            const maf_src_t& sr = op->src[i];
            MAF_REQUIRE(sr.ptr != nullptr, "conv: null source");
            MAF_REQUIRE(sr.C > 0 && sr.C % CH == 0 && sr.stride % CH == 0 && sr.coff % CH == 0, "conv: source C/stride/coff must be multiples of the 16-byte channel chunk");
            a.srcC[i] = sr.C;
            a.src[i] = sr.ptr; a.srcStride[i] = sr.stride; a.srcCoff[i] = sr.coff; a.srcMode[i] = sr.mode;
            csum += sr.C;
            if (sr.mode != MAF_SRC_DIRECT) any_special = true;
            if (sr.mode == MAF_SRC_UP2) MAF_REQUIRE(op->H % 2 == 0 && op->W % 2 == 0, "conv: UP2 source needs even H,W");
        }
        ksum += (i < op->nsrc) ? maf_cdiv(op->src[i].C, KS) : 0;
        a.cum[i + 1] = ksum;
    }
    MAF_REQUIRE(csum == op->Cin, "conv: sum of source channels != Cin");
    a.nsrc = op->nsrc;
    a.B = op->B; a.H = op->H; a.W = op->W; a.Hin = op->Hin; a.Win = op->Win;
    a.M = op->B * op->H * op->W;
    a.Cin = op->Cin; a.Cout = op->Cout;
    a.ksteps = ksum;   // every source padded to whole k-steps (matches the host weight packing)
    a.w = op->w; a.bias = op->bias; a.out = op->out;
    a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    MAF_REQUIRE(op->w && op->bias && op->out, "conv: null w/bias/out");
    a.twin = op->aux[0] != nullptr;
    if (a.twin) {
        MAF_REQUIRE(op->aux[1] && op->aux[2] && op->aux[3], "conv twin: aux = {src, w, bias, out} of the second conv, all four");
        MAF_REQUIRE(op->nsrc == 1 && op->src[0].mode != MAF_SRC_UP2, "conv twin: single-source variants only");
        MAF_REQUIRE(op->tile_k <= 2 || op->tile_k == 4 || op->tile_k == 8, "conv twin: generic, LDS-shared-weight and split-K variants only");
        a.src_t = op->aux[0]; a.w_t = op->aux[1]; a.bias_t = static_cast<const float*>(op->aux[2]); a.out_t = const_cast<void*>(op->aux[3]);
    }
    int pt = op->tile_p;
    const int ct = op->tile_c;
    MAF_REQUIRE(pt > 0 && ct > 0, "conv: tile_p/tile_c not set");
    const bool ks4 = op->tile_k == 4;
    MAF_REQUIRE((op->tile_k >= 0 && op->tile_k <= 5) || op->tile_k == 8, "conv: tile_k must be 1, 2 / 8 (weights through LDS), 3 / 5 (persistent streaming 1x1) or 4 (split-K)");
    const bool stream = op->tile_k == 3;
    const bool lb = op->tile_k == 2, dma = op->tile_k == 8;
    MAF_REQUIRE(!(lb || dma) || (op->dtype == MAF_F16 && !op->out_f32), "conv: tile_k = 2 / 8 are fp16-output variants");
    MAF_REQUIRE(!ks4 || pt == 1, "conv: split-K (tile_k = 4) needs tile_p = 1");
    a.nM = ks4 ? maf_cdiv(a.M, 16) : maf_cdiv(a.M, 64 * pt);
    if (ks4) pt = 0;                                                 // dispatch key of the split-K instantiations
    a.nN = maf_cdiv(op->Cout, 16 * ct);
    int var;
    if (op->kind == MAF_OP_CONV3X3S2_DGRAD) {
        MAF_REQUIRE(op->nsrc == 1 && op->src[0].mode == MAF_SRC_DIRECT && !a.twin, "conv3x3s2 dgrad: single direct source (dY)");
        MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->H - 1) / 2 + 1 == op->Hin && (op->W - 1) / 2 + 1 == op->Win, "conv3x3s2 dgrad: Hin,Win (the dY grid) must equal floor((H-1)/2)+1");
        MAF_REQUIRE(op->tile_k <= 1 && !op->out_f32 && op->act == MAF_ACT_NONE, "conv3x3s2 dgrad: generic variant, no epilogue");
        a.act = MAF_ACT_NONE;
        static const bool unsplit = getenv("MAF_DGRAD3_ALL_TAPS") != nullptr;          // A/B: every pixel walks all nine taps
        if (op->H % 2 == 0 && op->W % 2 == 0 && !unsplit) {                 // parity classes of equal size: each runs only its own taps
            a.dg_mc = op->B * (op->H / 2) * (op->W / 2);
            a.dg_nmc = maf_cdiv(a.dg_mc, 64 * pt);
            a.nM = 4 * a.dg_nmc;
        }
        return maf_conv_mfma_dgrad3(a, op->dtype, pt, ct, s);
    }
    if (op->kind == MAF_OP_CONV3X3S2) {
        MAF_REQUIRE(op->nsrc == 1 && op->src[0].mode == MAF_SRC_DIRECT, "conv3x3s2: single direct source");
        MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->Hin - 1) / 2 + 1 == op->H && (op->Win - 1) / 2 + 1 == op->W, "conv3x3s2: H,W must equal floor((Hin-1)/2)+1");
        var = VAR_3X3S2;
    } else if (op->nsrc == 1 && (op->src[0].mode == MAF_SRC_POOL2 || op->src[0].mode == MAF_SRC_SUB2)) {
        var = VAR_POOL2;
    } else if (op->nsrc == 1 && !any_special) {
        var = VAR_DIRECT;
    } else {
        for (int i = 0; i < op->nsrc; ++i) MAF_REQUIRE(op->src[i].mode != MAF_SRC_POOL2 && op->src[i].mode != MAF_SRC_SUB2, "conv1x1: POOL2 / SUB2 only as a single source");
        var = VAR_MULTI;
    }
    const bool outf32 = op->out_f32 != 0 && op->dtype == MAF_F16;
    MAF_REQUIRE(!outf32 || var == VAR_DIRECT, "conv: out_f32 only for single direct source");
    a.act = op->act;
    MAF_REQUIRE(op->act >= 0 && op->act <= 3, "conv: bad act");
    a.out_pairs = op->out_pairs;
    MAF_REQUIRE(!op->out_pairs || (op->kind == MAF_OP_CONV1X1 && op->tile_k == 5 && !a.twin && a.M % 2 == 0 && op->W % 2 == 0 && op->out_coff % 4 == 0 && op->out_stride % 4 == 0),
                "conv: out_pairs (pixel-pair output) is a tile_k = 5 conv1x1 option: even width, out_coff and out_stride multiples of 4");
    if (stream) {
        MAF_REQUIRE(op->dtype == MAF_F16 && !op->out_f32 && var == VAR_DIRECT && a.nsrc == 1, "conv: tile_k = 3 is an fp16 single-direct-source 1x1 variant");
        return maf_conv1x1_stream(a, pt, ct, s);
    }
    if (op->tile_k == 5) {
        MAF_REQUIRE(op->dtype == MAF_F16 && !op->out_f32 && (pt == 1 || pt == 2) && (var == VAR_DIRECT || var == VAR_MULTI || (var == VAR_POOL2 && op->kind == MAF_OP_CONV1X1)),
                    "conv: tile_k = 5 is an fp16 1x1 variant with tile_p = 1 (4 waves per workgroup) or 2 (8 waves)");
        a.nM = maf_cdiv(a.M, 16);
        a.stream_waves = pt == 2 ? 8 : 4;
        if (op->aux[2]) {
            // aux[2] = half of the scratch of the training-mode BatchNorm behind this conv, [reserved0 replicas][2][Cout] fp32: the epilogue adds the sum and the sum
            // of squares of the stored outputs (csrc/conv_stream_lds_st.hip); the BatchNorm call then runs with stats_ready = 1
            MAF_REQUIRE(var == VAR_DIRECT && op->src[0].mode == MAF_SRC_DIRECT && pt == 1 && op->act == MAF_ACT_NONE && !op->out_pairs && op->reserved0 >= 1 && op->reserved0 <= 64,
                        "conv: the statistics epilogue (aux[2]) needs one direct source, tile_p = 1, no activation, NHWC output, reserved0 = 1..64 replicas");
            a.stats = const_cast<float*>(static_cast<const float*>(op->aux[2])); a.stats_R = op->reserved0;
            return maf_conv1x1_stream_lds_st(a, ct, s);
        }
        return maf_conv1x1_stream_lds(a, var == VAR_POOL2 ? VAR_DIRECT : var, ct, s);      // a pooled / sub-sampled single source: the direct form with a.srcMode[0] set
    }
    if (lb) return maf_conv_mfma_f16_lb(a, var, pt, ct, s);
    if (dma) return maf_conv_mfma_f16_dma(a, var, pt, ct, s);
    if (op->dtype == MAF_F16) return maf_conv_mfma_f16(a, var, outf32, pt, ct, s);
    return maf_conv_mfma_f32(a, var, false, pt, ct, s);
}
