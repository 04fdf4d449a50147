// SPPF pooling pyramid: y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) with MaxPool2d(5, 1, 2), written
// straight into channel slices 1..3 of the 4*c_ concat buffer whose slice 0 (x) the preceding
// cv1 1x1 conv produced.  Replaces SPPF.m x3 + torch.cat of yolov6/layers/common.py:121-129.
//
// Chained 5x5 max pools with -inf padding equal clipped 5x5 / 9x9 / 13x13 window maxima, so one
// thread (one 16-byte channel group of one pixel) scans the 13x13 window once, row-wise, and
// tracks the three nested maxima (fallback for maps larger than 64x64).  The usual case — the 20x20
// P5 map at 640 px input — takes the LDS kernel below.
#include "maf_common.h"

namespace {

struct PoolArgs {
    const void* in; void* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff, CG;
};

template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void sppf_pool_kernel(const PoolArgs a) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.B * a.H * a.W * a.CG;
    if (tid >= total) return;
    const int cg = (int)(tid % a.CG);
    long long t = tid / a.CG;
    const int x = (int)(t % a.W); t /= a.W;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const T* in = static_cast<const T*>(a.in) + a.in_coff + cg * N;
    float m5[N], m9[N], m13[N];
#pragma unroll
    for (int j = 0; j < N; ++j) m5[j] = m9[j] = m13[j] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int iy = y + dy;
        if ((unsigned)iy >= (unsigned)a.H) continue;
        const T* row = in + (size_t)((size_t)b * a.H + iy) * a.W * a.in_stride;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int ix = x + dx;
            if ((unsigned)ix >= (unsigned)a.W) continue;
            const V v = *reinterpret_cast<const V*>(row + (size_t)ix * a.in_stride);
            const int adx = dx < 0 ? -dx : dx;
            const int r = ady > adx ? ady : adx;     // Chebyshev radius
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float f = (float)v[j];
                m13[j] = fmaxf(m13[j], f);
                if (r <= 4) m9[j] = fmaxf(m9[j], f);
                if (r <= 2) m5[j] = fmaxf(m5[j], f);
            }
        }
    }
    T* out = static_cast<T*>(a.out) + (size_t)(((size_t)b * a.H + y) * a.W + x) * a.out_stride + a.out_coff + cg * N;
    V o5, o9, o13;
#pragma unroll
    for (int j = 0; j < N; ++j) { o5[j] = (T)m5[j]; o9[j] = (T)m9[j]; o13[j] = (T)m13[j]; }
    *reinterpret_cast<V*>(out) = o5;
    *reinterpret_cast<V*>(out + a.C) = o9;
    *reinterpret_cast<V*>(out + 2 * a.C) = o13;
}

// LDS version for maps up to 64x64 (the 20x20 P5 map at 640 px): one workgroup = one image x one
// 16-byte channel group; the plane lives in LDS and each 5x5 pool is a separable row-max / col-max
// pass pair (5+5 reads instead of 25), chained three times.
template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(const PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    V* cur = reinterpret_cast<V*>(smem_raw);                 // [H*W]
    V* tmp = cur + a.H * a.W;                                // [H*W]
    // one image's channel groups on ONE XCD: a pixel's 16-byte groups share 128-byte lines, and workgroups go to the XCDs round-robin by block id — with
    // (channel group, image) = (blockIdx.x, blockIdx.y) every line was fetched by up to 8 L2s.  Block id -> XCD = id & 7; inside an XCD images in turn,
    // channel groups fastest (the grid is 1-D: CG * B blocks, B a multiple of 8 or not: the tail images keep the plain order)
    int b, cg;
    {
        const int bid = blockIdx.x, B8 = a.B & ~7;
        if (bid < B8 * a.CG) { const int xcd = bid & 7, j = bid >> 3; b = xcd + 8 * (j / a.CG); cg = j % a.CG; }
        else { const int r = bid - B8 * a.CG; b = B8 + r / a.CG; cg = r % a.CG; }
    }
    const int HW = a.H * a.W;
    const T* in = static_cast<const T*>(a.in) + (size_t)b * HW * a.in_stride + a.in_coff + cg * N;
    T* out = static_cast<T*>(a.out) + (size_t)b * HW * a.out_stride + a.out_coff + cg * N;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) cur[p] = *reinterpret_cast<const V*>(in + (size_t)p * a.in_stride);
    __syncthreads();
    for (int level = 0; level < 3; ++level) {
        for (int p = threadIdx.x; p < HW; p += blockDim.x) {            // row pass
            const int x = p % a.W, y = p / a.W;
            V m = cur[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                const int xx = x + d;
                if (d != 0 && (unsigned)xx < (unsigned)a.W) {
                    const V v = cur[y * a.W + xx];
#pragma unroll
                    for (int j = 0; j < N; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
                }
            }
            tmp[p] = m;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < HW; p += blockDim.x) {            // column pass
            const int x = p % a.W, y = p / a.W;
            V m = tmp[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                const int yy = y + d;
                if (d != 0 && (unsigned)yy < (unsigned)a.H) {
                    const V v = tmp[yy * a.W + x];
#pragma unroll
                    for (int j = 0; j < N; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
                }
            }
            *reinterpret_cast<V*>(out + (size_t)p * a.out_stride + level * a.C) = m;
            cur[p] = m;
        }
        __syncthreads();
    }
}

}  // namespace

int maf_launch_sppf_pool(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "sppf_pool: dtype must be f16/f32");
    const int N = op->dtype == MAF_F16 ? 8 : 4;
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && op->out, "sppf_pool: one direct source");
    MAF_REQUIRE(sr.C % N == 0 && sr.stride % N == 0 && sr.coff % N == 0 && op->out_stride % N == 0 && op->out_coff % N == 0, "sppf_pool: 16-byte channel alignment");
    PoolArgs a;
    a.in = sr.ptr; a.out = op->out; a.B = op->B; a.H = op->H; a.W = op->W; a.C = sr.C;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.CG = sr.C / N;
    const long long total = (long long)a.B * a.H * a.W * a.CG;
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    const size_t lds = (size_t)a.H * a.W * 16 * 2;
    if (lds <= 64 * 1024) {
        const dim3 gl(a.CG * a.B);
        if (op->dtype == MAF_F16) hipLaunchKernelGGL((sppf_pool_lds_kernel<half_t, half8_t, 8>), gl, b, lds, s, a);
        else hipLaunchKernelGGL((sppf_pool_lds_kernel<float, f32x4_t, 4>), gl, b, lds, s, a);
    } else if (op->dtype == MAF_F16) {
        hipLaunchKernelGGL((sppf_pool_kernel<half_t, half8_t, 8>), g, b, 0, s, a);
    } else {
        hipLaunchKernelGGL((sppf_pool_kernel<float, f32x4_t, 4>), g, b, 0, s, a);
    }
    return maf_check_hip(hipGetLastError(), "sppf_pool launch");
}
