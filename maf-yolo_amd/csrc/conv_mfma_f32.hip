// f32 instantiations of the MFMA conv kernels (see conv_mfma.inc.h)
#include "conv_mfma.inc.h"
int maf_conv_mfma_f32(const ConvArgs& a, int var, bool outf32, int pt, int ct, hipStream_t s) { return launch_var<float>(a, var, false, pt, ct, s); }
