// Instantiations of the persistent LDS-resident-weight 1x1 conv (conv_stream_lds.inc.h) for 26 .. 40 k-steps at tile_c = 4 (round 6) — the closing convs and the wide
// concats of MAF-YOLO-m / -s on the 20 x 20 ... 80 x 80 maps (832 ... 1280 input channels: RepHDW.conv2 of backbone.6 / .12 / .16 / .20 / .26 / .30, RepHDW.conv1 of
// .12 / .16 / .26 / .30, one_conv of backbone.8; yolov6/layers/common.py:928-946) that ran on the generic template because no instantiation existed: KS * 4 KiB of
// weight fragments <= 160 KiB of LDS, one workgroup of four waves per CU, a 64-channel tile per workgroup.  Chosen per layer by measurement (tuner.autotune).
#include "conv_stream_lds.inc.h"

namespace {

template <bool MULTI>
int launch_xwide(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
#define MAF_KS(K) case K: return launch_sl<4, K, MULTI>(a, s);
        MAF_KS(26) MAF_KS(28) MAF_KS(30) MAF_KS(32) MAF_KS(34) MAF_KS(36) MAF_KS(40)
#undef MAF_KS
    }
    maf_set_error("conv: tile_k = 5 with more than 24 k-steps exists for 26, 28, 30, 32, 34, 36, 40 k-steps at tile_c = 4");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

int maf_conv1x1_stream_lds_xwide(const ConvArgs& a, int var, hipStream_t s) {
    return var == VAR_MULTI ? launch_xwide<true>(a, s) : launch_xwide<false>(a, s);
}
