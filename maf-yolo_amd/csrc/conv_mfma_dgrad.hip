// Data gradient of the 3x3 stride-2 convs of the train-form graph (RepVGGBlock.rbr_dense common.py:219-224, ConvWrapper / Conv k = 3 s = 2
// :44-47; backward of yolov6/core/engine.py:164): the VAR_DGRAD3 instantiations of the MFMA conv template (see conv_mfma.inc.h).
#include "conv_mfma.inc.h"

namespace {
template <typename T>
int dgrad_tile(const ConvArgs& a, int pt, int ct, hipStream_t s) {
#define MAF_DG(P, C) if (pt == P && ct == C) return launch_act<T, P, C, VAR_DGRAD3, false>(a, s);
    MAF_DG(1, 2) MAF_DG(2, 2) MAF_DG(1, 4) MAF_DG(2, 4) MAF_DG(1, 8) MAF_DG(2, 8)
#undef MAF_DG
    maf_set_error("conv3x3s2 dgrad: tile_p in {1,2}, tile_c in {2,4,8}");
    return MAF_E_UNSUPPORTED;
}
}  // namespace

int maf_conv_mfma_dgrad3(const ConvArgs& a, int dtype, int pt, int ct, hipStream_t s) {
    return dtype == MAF_F16 ? dgrad_tile<half_t>(a, pt, ct, s) : dgrad_tile<float>(a, pt, ct, s);
}
