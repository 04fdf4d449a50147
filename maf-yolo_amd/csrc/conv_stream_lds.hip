// Persistent 1x1 convolution for longer reductions (up to 24 k-steps: Cin <= 768, KS * CT KiB of weights <= 160 KiB of LDS; single source or concat of up to 4 sources,
// incl. the nearest-x2 upsampled ones): the schedule of conv_stream.hip — a wave walks many 16-pixel tiles, the activation
// fragments of its next tile are in flight while the current tile is multiplied, activated and stored — with the weight
// fragments of the workgroup's channel tile resident in LDS for the whole kernel (KS * CT KiB, read with ds_read_b128) instead
// of registers.  Same reference code and operand layout as conv_mfma.inc.h (Conv.forward_fuse, torch.cat, nn.Upsample:
// yolov6/layers/common.py:49-50, 148-154; MAF-YOLO-n.yaml:21,26).
#include "conv_stream_lds.inc.h"

int maf_conv1x1_stream_lds_wide(const ConvArgs& a, int var, int ct, hipStream_t s);
int maf_conv1x1_stream_lds_w8(const ConvArgs& a, int var, int ct, hipStream_t s);

namespace {


template <int CT, bool MULTI>
int launch_sl_ks(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
#define MAF_KS(K) case K: if constexpr (K * CT <= 96) return launch_sl<CT, K, MULTI>(a, s); break;
        MAF_KS(2) MAF_KS(3) MAF_KS(4) MAF_KS(5) MAF_KS(6) MAF_KS(7) MAF_KS(8) MAF_KS(9) MAF_KS(10) MAF_KS(11) MAF_KS(12)
#undef MAF_KS
    }
    return maf_conv1x1_stream_lds_wide(a, MULTI ? VAR_MULTI : VAR_DIRECT, CT, s);     // longer reductions: conv_stream_lds_wide.hip

}

}  // namespace

int maf_conv1x1_stream_lds(const ConvArgs& a, int var, int ct, hipStream_t s) {
    if (a.stream_waves == 8) return maf_conv1x1_stream_lds_w8(a, var, ct, s);
    if (var == VAR_DIRECT) {
        if (ct == 2) return launch_sl_ks<2, false>(a, s);
        if (ct == 4) return launch_sl_ks<4, false>(a, s);
        if (ct == 6) return launch_sl_ks<6, false>(a, s);
        if (ct == 8) return launch_sl_ks<8, false>(a, s);
    } else if (var == VAR_MULTI) {
        if (ct == 4) return launch_sl_ks<4, true>(a, s);
        if (ct == 6) return launch_sl_ks<6, true>(a, s);
        if (ct == 8) return launch_sl_ks<8, true>(a, s);
    }
    maf_set_error("conv: tile_k = 5 supports direct / concat sources, tile_c in {2,4,6,8} (concat: {4,6,8})");
    return MAF_E_UNSUPPORTED;
}
