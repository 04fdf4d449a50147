// f16 instantiations of the MFMA conv kernel whose weight fragments reach the LDS by DMA, a ring of two-k-step stages (tile_k = 8; see conv_mfma.inc.h)
#include "conv_mfma.inc.h"
int maf_conv_mfma_f16_dma(const ConvArgs& a, int var, int pt, int ct, hipStream_t s) {
    if (var == VAR_DIRECT) return launch_dma_tile<VAR_DIRECT>(a, pt, ct, s);
    if (var == VAR_MULTI) return launch_dma_tile<VAR_MULTI>(a, pt, ct, s);
    if (var == VAR_3X3S2) return launch_dma_tile<VAR_3X3S2>(a, pt, ct, s);
    maf_set_error("conv: tile_k = 8 is not defined for a max-pooled source");
    return MAF_E_UNSUPPORTED;
}
