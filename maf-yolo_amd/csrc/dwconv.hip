// Depth-wise k x k (k in {3,5,7,9}) stride-1 "same" convolution + bias (+SiLU), NHWC.
//
// Replaces the merged DilatedReparamBlock / UniRepLKNetBlock of the deploy graph
// (yolov6/layers/common.py:3024-3026 deploy branch after merge_dilated_branches :3033-3051 and
// reparameterize :3085-3100) plus DepthBottleneckUni.act (common.py:922) where act = SiLU; the
// heads use act = none (common.py:1329,1333).
//
// HBM-bound stencil (4-40 FLOP/B): no MFMA.  One lane owns one 16-byte channel group
// (8 f16 / 4 f32 channels) and an R-pixel strip along x; consecutive lanes take consecutive channel
// groups, so every load/store instruction of a wave covers whole contiguous NHWC pixel rows.
// Row reuse: the (R+k-1) input vectors of a kernel row feed all R outputs from registers
// (k*(R+k-1)/R loads per output instead of k*k); vertical/neighbour-strip overlap is served by
// L1/L2 because a workgroup's threads are spatially compact (channel groups fastest, then x, then y).
// fp32 accumulation; weights [k*k][C] in the activation dtype.
#include "maf_common.h"

namespace {

struct DwArgs {
    const void* in; const void* w; const float* bias; void* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff, act;
    int CG;      // channel groups
    int XS;      // strips per row
};

template <typename T> struct Vec;
template <> struct Vec<half_t> {
    static constexpr int N = 8;
    typedef half8_t type;
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4_t type;
};

template <typename T, int K, int R>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwArgs a) {
    constexpr int N = Vec<T>::N;
    constexpr int P = K / 2;
    typedef typename Vec<T>::type vec_t;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.B * a.H * a.XS * a.CG;
    if (tid >= total) return;
    const int cg = (int)(tid % a.CG);
    long long t = tid / a.CG;
    const int xs = (int)(t % a.XS); t /= a.XS;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const int x0 = xs * R;
    const int c0 = cg * N;

    const T* in = static_cast<const T*>(a.in) + a.in_coff + c0;
    const T* w = static_cast<const T*>(a.w) + c0;
    float acc[R][N];
    {
        float bv[N];
#pragma unroll
        for (int j = 0; j < N; ++j) bv[j] = a.bias[c0 + j];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) acc[r][j] = bv[j];
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = y + ky - P;
        if ((unsigned)iy >= (unsigned)a.H) continue;      // zero padding: the row contributes nothing
        const T* row = in + (size_t)((size_t)b * a.H + iy) * a.W * a.in_stride;
        vec_t wv[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wv[kx] = *reinterpret_cast<const vec_t*>(w + (size_t)(ky * K + kx) * a.C);
#pragma unroll
        for (int i = 0; i < R + K - 1; ++i) {
            const int ix = x0 + i - P;
            if ((unsigned)ix >= (unsigned)a.W) continue;
            const vec_t v = *reinterpret_cast<const vec_t*>(row + (size_t)ix * a.in_stride);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kx = i - r;                     // compile-time after unrolling
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int j = 0; j < N; ++j) acc[r][j] = __builtin_fmaf((float)v[j], (float)wv[kx][j], acc[r][j]);
                }
            }
        }
    }
    T* out = static_cast<T*>(a.out) + a.out_coff + c0 + (size_t)((size_t)b * a.H + y) * a.W * a.out_stride;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (x0 + r >= a.W) break;
        vec_t o;
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = (T)maf_act_rt(acc[r][j], a.act);
        *reinterpret_cast<vec_t*>(out + (size_t)(x0 + r) * a.out_stride) = o;
    }
}

template <typename T>
int launch_t(const DwArgs& a, int k, hipStream_t s) {
    constexpr int R = 4;
    const long long total = (long long)a.B * a.H * a.XS * a.CG;
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    switch (k) {
        case 3: hipLaunchKernelGGL((dwconv_kernel<T, 3, R>), g, b, 0, s, a); break;
        case 5: hipLaunchKernelGGL((dwconv_kernel<T, 5, R>), g, b, 0, s, a); break;
        case 7: hipLaunchKernelGGL((dwconv_kernel<T, 7, R>), g, b, 0, s, a); break;
        case 9: hipLaunchKernelGGL((dwconv_kernel<T, 9, R>), g, b, 0, s, a); break;
        default: maf_set_error("dwconv: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
    return maf_check_hip(hipGetLastError(), "dwconv launch");
}

}  // namespace

int maf_launch_dwconv(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "dwconv: dtype must be f16/f32");
    const int N = op->dtype == MAF_F16 ? 8 : 4;
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr, "dwconv: one direct source");
    MAF_REQUIRE(op->Cin == op->Cout && sr.C == op->Cin && op->Cin % N == 0, "dwconv: C must be a multiple of the 16-byte channel group");
    MAF_REQUIRE(sr.stride % N == 0 && sr.coff % N == 0 && op->out_stride % N == 0 && op->out_coff % N == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->w && op->bias && op->out, "dwconv: null pointer");
    DwArgs a;
    a.in = sr.ptr; a.w = op->w; a.bias = op->bias; a.out = op->out;
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cin;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.act = op->act; a.CG = op->Cin / N; a.XS = maf_cdiv(op->W, 4);
    if (op->dtype == MAF_F16) return launch_t<half_t>(a, op->ksize, s);
    return launch_t<float>(a, op->ksize, s);
}
