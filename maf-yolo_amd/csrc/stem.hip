// Stem: backbone.0 RepVGGBlock in deploy form — 3x3 stride-2 pad-1 conv over the 3-channel NCHW
// image + bias + ReLU, writing NHWC.  Replaces rbr_reparam + nonlinearity of
// yolov6/layers/common.py:216-217 for the first layer (Cin = 3, MAF-YOLO-n.yaml:5) and, with
// in_dtype = MAF_U8, also folds the `imgs /= 255` pass of yolov6/core/evaler.py:161-163.
//
// K = 27 is too thin for MFMA and the layer is bound by its 2.4 MB/img read + 4.9 MB/img write, so
// this is a VALU direct conv: one thread owns 4 adjacent output pixels, reads the 9 input rows
// (3 channels x 3 ky) as one aligned 8-wide vector + 1 scalar each and keeps them in registers;
// weights (27 x Cout fp32) sit in LDS and are read as broadcast float4s shared by the 4 pixels;
// the results are staged in LDS and leave as fully coalesced 16-byte-per-lane stores over the workgroup's contiguous
// NHWC span (direct 16-byte pieces per (pixel, 8-channel pass) cost 3x write amplification in the PMC counters).
#include "maf_common.h"

namespace {

struct StemArgs {
    const void* img;   // [B,3,Hin,Win]
    const float* w;    // [27][Cout]
    const float* bias; // [Cout]
    void* out;
    int B, H, W, Hin, Win, Cout, out_stride, out_coff, act, staged;
    float in_scale;
};

// 8 consecutive input columns starting at an 8-element-aligned column, as floats
__device__ __forceinline__ void ld8(const half_t* p, float (&v)[8]) {
    const half8_t h = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void ld8(const uint8_t* p, float (&v)[8]) {
    const u32x2_t w = *reinterpret_cast<const u32x2_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (float)((w[0] >> (8 * i)) & 0xffu); v[4 + i] = (float)((w[1] >> (8 * i)) & 0xffu); }
}

// One thread = 4 horizontally adjacent output pixels (input columns 2x-1 .. 2x+7: one aligned 8-wide
// vector load + one scalar per (channel, ky)), all Cout channels in passes of 8.
template <typename TI, typename TO, bool STAGED>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [27][Cout] + [Cout]
    const int nW = 27 * a.Cout;
    for (int i = threadIdx.x; i < nW + a.Cout; i += blockDim.x) smem[i] = i < nW ? a.w[i] : a.bias[i - nW];
    __syncthreads();
    // weights / bias are wave-uniform: they come through the scalar cache into SGPRs (s_load), not LDS/VGPRs
    const int XQ = a.W >> 2;                                      // quads per output row (W % 4 == 0)
    const int total = a.B * a.H * XQ;
    const int tq = blockIdx.x * blockDim.x + threadIdx.x;
    if (!STAGED && tq >= total) return;
    const int t = tq < total ? tq : total - 1;
    // staging area: one slot of 4 pixels x Cout outputs (+16 B pad: conflict-free 16-byte LDS stores) per thread
    const int cpt = (4 * a.Cout * (int)sizeof(TO)) >> 4;          // 16-byte chunks per thread
    char* stage = reinterpret_cast<char*>(smem + nW + a.Cout) + (size_t)threadIdx.x * (cpt + 1) * 16;
    const int xq = t % XQ, t2 = t / XQ, y = t2 % a.H, b = t2 / a.H;
    const int x0 = xq * 4;
    const TI* img = static_cast<const TI*>(a.img) + (size_t)b * 3 * a.Hin * a.Win;
    f32x2_t in[9][5];                                              // [c*3+ky][column 2x0-1 .. 2x0+7, in register pairs]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * y - 1 + ky;
            const bool rok = (unsigned)iy < (unsigned)a.Hin;
            const TI* row = img + ((size_t)c * a.Hin + (rok ? iy : 0)) * a.Win + 2 * x0;
            float v[8];
            ld8(row, v);
            const float left = x0 > 0 ? (float)row[-1] : 0.f;
            const float m = rok ? a.in_scale : 0.f;
            in[c * 3 + ky][0][0] = left * m;
            in[c * 3 + ky][4][1] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) in[c * 3 + ky][(1 + i) >> 1][(1 + i) & 1] = v[i] * m;
        }
    TO* o = static_cast<TO*>(a.out) + ((size_t)(b * a.H + y) * a.W + x0) * a.out_stride + a.out_coff;
#pragma unroll 1
    for (int c0 = 0; c0 < a.Cout; c0 += 8) {
        f32x2_t acc[4][4];                                         // v_pk_fma_f32: two fp32 FMAs per lane per issue
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[p][j] = f32x2_t{smem[nW + c0 + 2 * j], smem[nW + c0 + 2 * j + 1]};
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            // keep one (channel, ky) row of weights (6 LDS reads = 24 VGPRs) in flight: without the fence the scheduler
            // hoists all 54 reads of the pass (216 VGPRs => 1 wave/SIMD)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int k = r * 3 + kx;
                f32x2_t wk[4];
                {
                    const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0]);
                    const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0 + 4]);
                    wk[0] = f32x2_t{w0[0], w0[1]}; wk[1] = f32x2_t{w0[2], w0[3]};
                    wk[2] = f32x2_t{w1[0], w1[1]}; wk[3] = f32x2_t{w1[2], w1[3]};
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2_t vv = in[r][(2 * p + kx) >> 1];     // the column is the low (kx even) or high (kx odd) register of the pair
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (kx & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[p][j]) : "v"(vv), "v"(wk[j]));
                        else        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[p][j]) : "v"(vv), "v"(wk[j]));
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (sizeof(TO) == 2) {
                half8_t v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (half_t)fmaxf(acc[p][j >> 1][j & 1], 0.f);
                if (STAGED) *reinterpret_cast<half8_t*>(stage + ((size_t)p * a.Cout + c0) * 2) = v;
                else *reinterpret_cast<half8_t*>(reinterpret_cast<half_t*>(o) + (size_t)p * a.out_stride + c0) = v;
            } else {
                f32x4_t v0, v1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(acc[p][j >> 1][j & 1], 0.f); v1[j] = fmaxf(acc[p][2 + (j >> 1)][j & 1], 0.f); }
                float* of = STAGED ? reinterpret_cast<float*>(stage) + (size_t)p * a.Cout + c0
                                     : reinterpret_cast<float*>(o) + (size_t)p * a.out_stride + c0;
                *reinterpret_cast<f32x4_t*>(of) = v0;
                *reinterpret_cast<f32x4_t*>(of + 4) = v1;
            }
        }
    }
    if (!STAGED) return;
    __syncthreads();
    // thread t of the grid owns pixels 4t .. 4t+3, so the workgroup's output is ONE contiguous span (out_stride == Cout)
    const int first = blockIdx.x * blockDim.x;
    const int nthr = total - first < (int)blockDim.x ? total - first : (int)blockDim.x;
    const char* sbase = reinterpret_cast<const char*>(smem + nW + a.Cout);
    char* gbase = reinterpret_cast<char*>(static_cast<TO*>(a.out) + (size_t)first * 4 * a.Cout);
    for (int q = threadIdx.x; q < nthr * cpt; q += blockDim.x) {
        const int thr = q / cpt, off = q - thr * cpt;
        *reinterpret_cast<f32x4_t*>(gbase + (size_t)q * 16) = *reinterpret_cast<const f32x4_t*>(sbase + ((size_t)thr * (cpt + 1) + off) * 16);
    }
}

}  // namespace

int maf_launch_stem(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "stem: dtype must be f16/f32");
    MAF_REQUIRE(op->Cin == 3 && op->Cout % 8 == 0 && op->Cout <= 256, "stem: Cin must be 3, Cout a multiple of 8");
    MAF_REQUIRE(op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "stem: out stride/coff multiples of 8");
    MAF_REQUIRE(op->Win % 8 == 0, "stem: image width must be a multiple of 8");
    MAF_REQUIRE(op->act == MAF_ACT_RELU, "stem: the RepVGG stem ends in ReLU (common.py:198)");
    MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->Hin - 1) / 2 + 1 == op->H && (op->Win - 1) / 2 + 1 == op->W, "stem: H,W must equal floor((Hin-1)/2)+1");
    MAF_REQUIRE(op->src[0].ptr && op->w && op->bias && op->out, "stem: null pointer");
    StemArgs a;
    a.img = op->src[0].ptr; a.w = static_cast<const float*>(op->w); a.bias = op->bias; a.out = op->out;
    a.B = op->B; a.H = op->H; a.W = op->W; a.Hin = op->Hin; a.Win = op->Win; a.Cout = op->Cout;
    a.out_stride = op->out_stride; a.out_coff = op->out_coff; a.act = op->act;
    a.in_scale = op->in_dtype == MAF_U8 ? 1.0f / 255.0f : 1.0f;
    const int M = op->B * op->H * (op->W / 4);
    const int es = op->dtype == MAF_F16 ? 2 : 4;
    a.staged = op->out_stride == op->Cout && op->out_coff == 0;     // dense output: coalesce through LDS
    const int slot = 4 * op->Cout * es + 16;
    int threads = 256;
    if (a.staged) { threads = (44 * 1024 / slot) / 64 * 64; threads = threads > 256 ? 256 : threads < 64 ? 64 : threads; }
    const dim3 g(maf_cdiv(M, threads)), b(threads);
    const size_t sh = (size_t)(28 * op->Cout) * sizeof(float) + (a.staged ? (size_t)threads * slot : 0);
    MAF_REQUIRE(sh <= 64 * 1024, "stem: Cout too large for the LDS staging area");
#define MAF_STEM(TI, TO) do { if (a.staged) hipLaunchKernelGGL((stem_kernel<TI, TO, true>), g, b, sh, s, a); else hipLaunchKernelGGL((stem_kernel<TI, TO, false>), g, b, sh, s, a); } while (0)
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
