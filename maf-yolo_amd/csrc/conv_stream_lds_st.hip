// The persistent LDS-resident-weight 1x1 conv (conv_stream_lds.inc.h) with the statistics epilogue of the training-mode BatchNorm behind it (ST = true):
// Conv = conv -> BatchNorm2d -> SiLU of yolov6/layers/common.py:29-47 in training mode (Trainer.train_in_steps, yolov6/core/engine.py:141-167) normalises with
// the batch statistics of the conv's output; the statistics pass of maf_bn_forward (csrc/bn_act.hip) read that output once more, one launch per layer.  Here the
// conv's own epilogue accumulates them (VERDICT r4 #1b).  Single direct source, four waves, ksteps <= 12 and ksteps * tile_c <= 96 — the instantiations the
// training step's tuner picks for the 1x1 convs of MAF-YOLO-n / s / m whose K is at most 384.
#include "conv_stream_lds.inc.h"

namespace {

template <int CT>
int launch_st_ks(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
#define MAF_KS(K) case K: if constexpr (K * CT <= 96) return launch_sl<CT, K, false, 4, true>(a, s); break;
        MAF_KS(2) MAF_KS(3) MAF_KS(4) MAF_KS(5) MAF_KS(6) MAF_KS(7) MAF_KS(8) MAF_KS(9) MAF_KS(10) MAF_KS(11) MAF_KS(12)
#undef MAF_KS
    }
    maf_set_error("conv: the statistics epilogue (aux[2]) exists for 2 <= K / 32 <= 12 and K / 32 * tile_c <= 96");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

extern "C" int maf_conv1x1_stats_supported(int32_t ksteps, int32_t tile_c) {
    return (tile_c == 2 || tile_c == 4 || tile_c == 6 || tile_c == 8) && ksteps >= 2 && ksteps <= 12 && ksteps * tile_c <= 96;
}

int maf_conv1x1_stream_lds_st(const ConvArgs& a, int ct, hipStream_t s) {
    if (ct == 2) return launch_st_ks<2>(a, s);
    if (ct == 4) return launch_st_ks<4>(a, s);
    if (ct == 6) return launch_st_ks<6>(a, s);
    if (ct == 8) return launch_st_ks<8>(a, s);
    maf_set_error("conv: the statistics epilogue needs tile_c in {2, 4, 6, 8}");
    return MAF_E_UNSUPPORTED;
}
