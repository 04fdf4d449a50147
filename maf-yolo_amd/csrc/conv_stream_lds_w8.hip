// The persistent LDS-resident-weight 1x1 conv (conv_stream_lds.inc.h) with EIGHT waves per workgroup (MAF_OP_CONV1X1, tile_k = 5, tile_p = 2) for the
// instantiations whose weights take 64 KiB of LDS or more — the K-heavy layers of the 20 x 20 / 40 x 40 maps (Conv.forward_fuse over the wide MAFPN concats,
// yolov6/layers/common.py:49-50, configs/yaml/MAF-YOLO-n.yaml:16-42), where the LDS leaves one or two workgroups per CU: with four waves each that is one or
// two waves per SIMD, each with ONE activation tile in flight ahead of the one it multiplies — the kernel waits on L2 latency.  Eight waves behind the same
// copy of the weights double the tiles in flight per CU and halve the tiles a wave walks.  A tuner candidate beside the four-wave form (engine.py:autotune).
#include "conv_stream_lds.inc.h"

namespace {

template <int CT, bool MULTI>
int launch_w8(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
#define MAF_KS(K) case K: if constexpr (K * CT >= 64 && K * CT <= 160) return launch_sl<CT, K, MULTI, 8>(a, s); break;
        MAF_KS(8) MAF_KS(9) MAF_KS(10) MAF_KS(11) MAF_KS(12) MAF_KS(13) MAF_KS(14) MAF_KS(15) MAF_KS(16) MAF_KS(17) MAF_KS(18) MAF_KS(19) MAF_KS(20) MAF_KS(24)
#undef MAF_KS
    }
    maf_set_error("conv: tile_k = 5 with tile_p = 2 (8 waves per workgroup) needs 64 <= ksteps * tile_c <= 160 and ksteps in 8..20 or 24");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

int maf_conv1x1_stream_lds_w8(const ConvArgs& a, int var, int ct, hipStream_t s) {
    if (ct == 4) return var == VAR_MULTI ? launch_w8<4, true>(a, s) : launch_w8<4, false>(a, s);
    if (ct == 6) return var == VAR_MULTI ? launch_w8<6, true>(a, s) : launch_w8<6, false>(a, s);
    if (ct == 8) return var == VAR_MULTI ? launch_w8<8, true>(a, s) : launch_w8<8, false>(a, s);
    maf_set_error("conv: tile_k = 5 with tile_p = 2 supports tile_c in {4, 6, 8}");
    return MAF_E_UNSUPPORTED;
}
