// Stem: backbone.0 RepVGGBlock in deploy form — 3x3 stride-2 pad-1 conv over the 3-channel NCHW
// image + bias + ReLU, writing NHWC.  Replaces rbr_reparam + nonlinearity of
// yolov6/layers/common.py:216-217 for the first layer (Cin = 3, MAF-YOLO-n.yaml:5) and, with
// in_dtype = MAF_U8, also folds the `imgs /= 255` pass of yolov6/core/evaler.py:161-163.
//
// K = 27 is too thin for MFMA and the layer is bound by its 2.4 MB/img read + 4.9 MB/img write, so
// this is a VALU direct conv: one thread per output pixel keeps its 27 taps in registers, weights
// (27 x Cout fp32) sit in LDS and are read as broadcast float4s; each thread emits its Cout
// channels as 16-byte NHWC stores (adjacent lanes = adjacent pixels => fully coalesced rows).
#include "maf_common.h"

namespace {

struct StemArgs {
    const void* img;   // [B,3,Hin,Win]
    const float* w;    // [27][Cout]
    const float* bias; // [Cout]
    void* out;
    int B, H, W, Hin, Win, Cout, out_stride, out_coff, act;
    float in_scale;
};

template <typename TI> __device__ __forceinline__ float ld_in(const TI* p) { return (float)*p; }

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [27][Cout] + [Cout]
    const int nW = 27 * a.Cout;
    for (int i = threadIdx.x; i < nW + a.Cout; i += blockDim.x) smem[i] = i < nW ? a.w[i] : a.bias[i - nW];
    __syncthreads();
    const int M = a.B * a.H * a.W;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int x = m % a.W, t = m / a.W, y = t % a.H, b = t / a.H;
    const TI* img = static_cast<const TI*>(a.img) + (size_t)b * 3 * a.Hin * a.Win;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
                const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
                in[(c * 3 + ky) * 3 + kx] = ok ? ld_in<TI>(img + ((size_t)c * a.Hin + iy) * a.Win + ix) * a.in_scale : 0.f;
            }
    TO* o = static_cast<TO*>(a.out) + (size_t)m * a.out_stride + a.out_coff;
    for (int c0 = 0; c0 < a.Cout; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = smem[nW + c0 + j];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0]);
            const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0 + 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_fmaf(in[k], w0[j], acc[j]);
                acc[4 + j] = __builtin_fmaf(in[k], w1[j], acc[4 + j]);
            }
        }
        if (sizeof(TO) == 2) {
            half8_t v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)maf_act_rt(acc[j], a.act);
            *reinterpret_cast<half8_t*>(reinterpret_cast<half_t*>(o) + c0) = v;
        } else {
            f32x4_t v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = maf_act_rt(acc[j], a.act); v1[j] = maf_act_rt(acc[4 + j], a.act); }
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(o) + c0) = v0;
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(o) + c0 + 4) = v1;
        }
    }
}

}  // namespace

int maf_launch_stem(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "stem: dtype must be f16/f32");
    MAF_REQUIRE(op->Cin == 3 && op->Cout % 8 == 0 && op->Cout <= 256, "stem: Cin must be 3, Cout a multiple of 8");
    MAF_REQUIRE(op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "stem: out stride/coff multiples of 8");
    MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->Hin - 1) / 2 + 1 == op->H && (op->Win - 1) / 2 + 1 == op->W, "stem: H,W must equal floor((Hin-1)/2)+1");
    MAF_REQUIRE(op->src[0].ptr && op->w && op->bias && op->out, "stem: null pointer");
    StemArgs a;
    a.img = op->src[0].ptr; a.w = static_cast<const float*>(op->w); a.bias = op->bias; a.out = op->out;
    a.B = op->B; a.H = op->H; a.W = op->W; a.Hin = op->Hin; a.Win = op->Win; a.Cout = op->Cout;
    a.out_stride = op->out_stride; a.out_coff = op->out_coff; a.act = op->act;
    a.in_scale = op->in_dtype == MAF_U8 ? 1.0f / 255.0f : 1.0f;
    const int M = op->B * op->H * op->W;
    const dim3 g(maf_cdiv(M, 256)), b(256);
    const size_t sh = (size_t)(28 * op->Cout) * sizeof(float);
#define MAF_STEM(TI, TO) hipLaunchKernelGGL((stem_kernel<TI, TO>), g, b, sh, s, a)
    if (op->dtype == MAF_F16) {
        if (op->in_dtype == MAF_F16) MAF_STEM(half_t, half_t);
        else if (op->in_dtype == MAF_F32) MAF_STEM(float, half_t);
        else if (op->in_dtype == MAF_U8) MAF_STEM(uint8_t, half_t);
        else { maf_set_error("stem: bad in_dtype"); return MAF_E_ARG; }
    } else {
        if (op->in_dtype == MAF_F16) MAF_STEM(half_t, float);
        else if (op->in_dtype == MAF_F32) MAF_STEM(float, float);
        else if (op->in_dtype == MAF_U8) MAF_STEM(uint8_t, float);
        else { maf_set_error("stem: bad in_dtype"); return MAF_E_ARG; }
    }
#undef MAF_STEM
    return maf_check_hip(hipGetLastError(), "stem launch");
}
