// Instantiations of the persistent LDS-resident-weight 1x1 conv (conv_stream_lds.inc.h) for 13 .. 24 k-steps — the wide concats of the MAFPN neck on
// the 40 x 40 / 20 x 20 maps (448, 480, 576, 768 channels of MAF-YOLO-n; 416 .. 640 of s): KS * CT KiB of weight fragments <= 160 KiB of LDS, i.e. one
// workgroup per CU, one wave per SIMD with the whole register file for its two activation tiles.  tile_c in {4, 6, 8}.
#include "conv_stream_lds.inc.h"

int maf_conv1x1_stream_lds_xwide(const ConvArgs& a, int var, hipStream_t s);      // 26 .. 40 k-steps at tile_c = 4 (conv_stream_lds_xwide.hip)

namespace {

template <int CT, bool MULTI>
int launch_wide(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
#define MAF_KS(K) case K: if constexpr (K * CT <= 160) return launch_sl<CT, K, MULTI>(a, s); break;
        MAF_KS(13) MAF_KS(14) MAF_KS(15) MAF_KS(16) MAF_KS(17) MAF_KS(18) MAF_KS(19) MAF_KS(20) MAF_KS(24)
#undef MAF_KS
    }
    maf_set_error("conv: tile_k = 5 (persistent, LDS-resident weights) needs ksteps in 2..20 or 24 and ksteps * tile_c <= 160 (<= 96 up to 12 k-steps)");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

int maf_conv1x1_stream_lds_wide(const ConvArgs& a, int var, int ct, hipStream_t s) {
    if (ct == 4 && a.ksteps > 24) return maf_conv1x1_stream_lds_xwide(a, var, s);
    if (ct == 4) return var == VAR_MULTI ? launch_wide<4, true>(a, s) : launch_wide<4, false>(a, s);
    if (ct == 6) return var == VAR_MULTI ? launch_wide<6, true>(a, s) : launch_wide<6, false>(a, s);
    if (ct == 8) return var == VAR_MULTI ? launch_wide<8, true>(a, s) : launch_wide<8, false>(a, s);
    maf_set_error("conv: tile_k = 5 with more than 12 k-steps supports tile_c in {4, 6, 8}");
    return MAF_E_UNSUPPORTED;
}
