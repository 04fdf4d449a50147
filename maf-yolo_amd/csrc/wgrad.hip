// Weight gradient of the dense convolutions of the train-form graph:  dW[co][ci] = sum over pixels m of dY[m][co] * X[m][ci].
//
// Backward of nn.Conv2d(k=1) inside Conv / Head_DepthUni (yolov6/layers/common.py:29-50, 1331-1335) as the reference's
// autograd computes it in Trainer.train_in_steps (yolov6/core/engine.py:152-160).  As a GEMM it has a tiny output
// (Cout x Cin <= 576 x 576) and a reduction over up to 819 200 pixels; vendor TN GEMMs spend ~0.5 ms on it.  Here the
// pixel range is cut into chunks (one workgroup each), a chunk is walked 64 pixels at a time, every wave owns a block of
// 16 x 16 output tiles whose accumulators stay in registers for the whole chunk, and the partial dW of the chunk is added
// to the fp32 result with atomics.  fp16 operands, fp32 accumulation.
//
// The same kernel is the weight gradient of the stride-2 convs (RepVGGBlock.rbr_dense / rbr_1x1 common.py:202-203, ConvWrapper :76-83):
// blockIdx.z walks the taps, and the X tile of tap (ky, kx) is gathered from pixel (2 oy - 1 + ky, 2 ox - 1 + kx) of the full-resolution
// input (zeros outside the image) while it is staged — no im2col tensor, no per-tap copies.  The 3x3 result is TAP-MAJOR,
// [ky][kx][Cout][Cin]: the 16 lanes of an atomic instruction then hit one 64-byte line instead of 16 lines 36 bytes apart (the
// [Cout][Cin][3][3] layout made the atomics, not the streaming, the bound: 7 us per pixel chunk on 192 -> 192).  Inputs wider than 256
// channels are cut into channel chunks by the host wrapper (the LDS tile holds one chunk).
//
// NO transposing stores: both tiles stay in their NHWC orientation in LDS ([pixel][channel], 16-byte stores of what the 16-byte
// global loads returned) and the MFMA operands — 8 consecutive k = pixels of ONE channel per lane — come out of gfx950's transposing
// LDS read: ds_read_b64_tr_b16, where lane p of a 16-lane group passes the address of (row k0 + (p >> 2), columns 4 (p & 3) ..) and
// receives (rows k0 .. k0 + 3, column p)  [tools/tr_probe.py].  (The first version transposed while staging — eight 2-byte LDS
// stores per 16 bytes loaded — and ran at half the speed on every shape.)  The wave grid is WI x WJ (WI * WJ = 4): wave (wi, wj) owns
// the TI x NJ block of output tiles at (co tile wi*TI, ci tile wj*NJ), so one k-slab costs TI + NJ operand fetches for TI * NJ MFMAs;
// the tiles of pixel step s + 1 are loaded into registers before the MFMAs of step s and written to LDS after them.
#include <cmath>
#include <cstdlib>
#include "maf_common.h"

namespace {

constexpr int kPix = 64;              // pixels per staging step

struct WgArgs2 {
    const half_t* x; const half_t* dy; float* dw;
    int M, Cin, Cout, x_stride, dy_stride;
    int chunk;            // pixels per workgroup (multiple of 64)
    int WJ;               // waves along ci (1, 2, 4); WI = 4 / WJ along co
    int gather, Ho, Wo, Hs, Ws, tap0;
    int dw_stride;        // floats between consecutive co rows of one tap
    long long dw_tap;     // floats between consecutive taps (3x3: tap-major result)
    int ntaps, cchunk, Cin_all;   // blockIdx.z = channel chunk * ntaps + tap: chunk z / ntaps covers input channels [chunk * cchunk, min(Cin_all, (chunk + 1) * cchunk))
};

typedef __fp16 tr4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) tr4_t* lds_tr_ptr;

__device__ __forceinline__ half8_t tr_frag(const half_t* lo, const half_t* hi) {          // k0 .. k0+3 | k0+4 .. k0+7 of the lane's channel
    const tr4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(lo));
    const tr4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(hi));
    typedef __fp16 tr8_t __attribute__((__vector_size__(8 * sizeof(__fp16))));
    const tr8_t v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(half8_t, v);
}

template <int TI, int NJ>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const WgArgs2 a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int XR = 2 * NJ, DR = 2 * TI;                               // 16-byte chunks a thread stages per step (upper bounds)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int WJ = a.WJ, WI = 4 / WJ, wj = wave % WJ, wi = wave / WJ;
    const int cib = WJ * NJ * 16, cob = WI * TI * 16;                    // channels of X / dY staged per step
    const int SX = cib + 8, SD = cob + 8;                                // LDS row strides in halfs
    const int co0 = blockIdx.y * cob;
    half_t* Xs = reinterpret_cast<half_t*>(smem_raw);                    // [kPix][SX]
    half_t* Ds = Xs + kPix * SX;                                         // [kPix][SD]
    const int zc = blockIdx.z / a.ntaps, zt = blockIdx.z - zc * a.ntaps;
    const int tap = a.tap0 + zt, tky = tap / 3, tkx = tap - tky * 3;
    // channel chunk of this workgroup (inputs wider than 256 channels: the LDS tile holds one chunk; all chunks are ONE grid — as separate launches of ~150 workgroups each
    // the 20 x 20 layers ran their three chunks one after the other on a chip they could not fill)
    const half_t* xbase = a.x + zc * a.cchunk;
    float* dwbase = a.dw + zc * a.cchunk;
    const int Cin = min(a.cchunk, a.Cin_all - zc * a.cchunk);

    // staging items of this thread: (row, 8-channel chunk), the same for every step
    const int xg = cib >> 3, dg = cob >> 3;
    int xrow[XR], xch[XR], drow[DR], dch[DR];
#pragma unroll
    for (int u = 0; u < XR; ++u) { const int it = tid + 256 * u; xrow[u] = it / xg; xch[u] = (it - xrow[u] * xg) * 8; if (xrow[u] >= kPix) xrow[u] = -1; }
#pragma unroll
    for (int u = 0; u < DR; ++u) { const int it = tid + 256 * u; drow[u] = it / dg; dch[u] = (it - drow[u] * dg) * 8; if (drow[u] >= kPix) drow[u] = -1; }

    f32x4_t acc[TI][NJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t)0.f;

    const int m_begin = blockIdx.x * a.chunk, m_end = min(a.M, m_begin + a.chunk);
    half8_t xr[XR], dr[DR];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int m = m0 + xrow[u];
            if (xrow[u] >= 0 && m < m_end && xch[u] < Cin) {
                if (!a.gather) {
                    v = *reinterpret_cast<const half8_t*>(xbase + (size_t)m * a.x_stride + xch[u]);
                } else {
                    const int ox = m % a.Wo, t2 = m / a.Wo, oy = t2 % a.Ho, bb = t2 / a.Ho;
                    const int iy = 2 * oy - 1 + tky, ix = 2 * ox - 1 + tkx;
                    if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws)
                        v = *reinterpret_cast<const half8_t*>(xbase + ((size_t)(bb * a.Hs + iy) * a.Ws + ix) * a.x_stride + xch[u]);
                }
            }
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int m = m0 + drow[u];
            if (drow[u] >= 0 && m < m_end && co0 + dch[u] < a.Cout) v = *reinterpret_cast<const half8_t*>(a.dy + (size_t)m * a.dy_stride + co0 + dch[u]);
            dr[u] = v;
        }
    };
    // per-lane operand addresses: row g*8 + (p >> 2), column 4 (p & 3) of the wave's first tile
    const half_t* abase = Ds + (g * 8 + (p >> 2)) * SD + wi * TI * 16 + (p & 3) * 4;
    const half_t* bbase = Xs + (g * 8 + (p >> 2)) * SX + wj * NJ * 16 + (p & 3) * 4;

    fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += kPix) {
        __syncthreads();                                                 // the MFMAs of the previous step have read the tiles
#pragma unroll
        for (int u = 0; u < XR; ++u)
            if (xrow[u] >= 0) *reinterpret_cast<half8_t*>(Xs + xrow[u] * SX + xch[u]) = xr[u];
#pragma unroll
        for (int u = 0; u < DR; ++u)
            if (drow[u] >= 0) *reinterpret_cast<half8_t*>(Ds + drow[u] * SD + dch[u]) = dr[u];
        __syncthreads();
        if (m0 + kPix < m_end) fetch(m0 + kPix);                         // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < kPix / 32; ++ks) {
            half8_t av[TI], bv[NJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) av[i] = tr_frag(abase + (ks * 32) * SD + i * 16, abase + (ks * 32 + 4) * SD + i * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = tr_frag(bbase + (ks * 32) * SX + j * 16, bbase + (ks * 32 + 4) * SX + j * 16);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- accumulator lane (g, p): rows co = 4g + r, column ci = p of its tile
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = (wj * NJ + j) * 16 + p;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wi * TI + i) * 16 + g * 4 + r;
                if (co < a.Cout && ci < Cin) atomicAdd(dwbase + (size_t)(tap - a.tap0) * a.dw_tap + (size_t)co * a.dw_stride + ci, acc[i][j][r]);
            }
        }
}

// NINE TAPS IN ONE WORKGROUP, for the narrowest 3x3 layer (the first stem conv, 8 -> 24 channels on 640^2: one input-channel tile).  With one tap per workgroup dY — the big operand there, 32 x 320 x 320 x 24 =
// 157 MB — is read nine times (1.4 GB, 300 us per launch); here a pixel step stages dY once and the nine gathered X tiles next to it, and
// the (tap, co tile, ci tile) output tiles are dealt round-robin to the four waves.
template <int TPW>
__global__ __launch_bounds__(256) void wgrad_taps_kernel(const WgArgs2 a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int tci = (a.Cin + 15) >> 4, tco = (a.Cout + 15) >> 4;
    const int cib = tci * 16, cob = tco * 16;
    const int SX = cib + 8, SD = cob + 8;                                // LDS row strides in halfs
    half_t* Xs = reinterpret_cast<half_t*>(smem_raw);                    // [9][kPix][SX]
    half_t* Ds = Xs + 9 * kPix * SX;                                     // [kPix][SD]
    const int ntile1 = tci * tco, ntile = 9 * ntile1;                    // virtual tile vt = (tap * tco + ti) * tci + tj, wave = vt % 4

    constexpr int XR = 10, DR = 2;                                       // 16-byte chunks a thread stages per step (9 * 64 * cib / 8 / 256 <= 9 for cib = 32)
    const int xg = cib >> 3, dg = cob >> 3;
    const int nx = 9 * kPix * xg, nd = kPix * dg;

    f32x4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (f32x4_t)0.f;

    const int m_begin = blockIdx.x * a.chunk, m_end = min(a.M, m_begin + a.chunk);
    half8_t xr[XR], dr[DR];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int it = tid + 256 * u;
            if (it < nx) {
                const int ch = (it % xg) * 8, r2 = it / xg;
                const int row = r2 % kPix, tap = r2 / kPix;
                const int m = m0 + row;
                if (m < m_end && ch < a.Cin) {
                    const int ox = m % a.Wo, t2 = m / a.Wo, oy = t2 % a.Ho, bb = t2 / a.Ho;
                    const int tky = tap / 3, tkx = tap - tky * 3;
                    const int iy = 2 * oy - 1 + tky, ix = 2 * ox - 1 + tkx;
                    if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws)
                        v = *reinterpret_cast<const half8_t*>(a.x + ((size_t)(bb * a.Hs + iy) * a.Ws + ix) * a.x_stride + ch);
                }
            }
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int it = tid + 256 * u;
            if (it < nd) {
                const int ch = (it % dg) * 8, row = it / dg;
                const int m = m0 + row;
                if (m < m_end && ch < a.Cout) v = *reinterpret_cast<const half8_t*>(a.dy + (size_t)m * a.dy_stride + ch);
            }
            dr[u] = v;
        }
    };
    fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += kPix) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            const int it = tid + 256 * u;
            if (it < nx) {
                const int ch = (it % xg) * 8, r2 = it / xg;
                *reinterpret_cast<half8_t*>(Xs + (size_t)r2 * SX + ch) = xr[u];          // r2 = tap * kPix + row
            }
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const int it = tid + 256 * u;
            if (it < nd) *reinterpret_cast<half8_t*>(Ds + (it / dg) * SD + (it % dg) * 8) = dr[u];
        }
        __syncthreads();
        if (m0 + kPix < m_end) fetch(m0 + kPix);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int vt = wave + 4 * t;
            if (vt < ntile) {
                const int tap = vt / ntile1, r1 = vt - tap * ntile1;
                const int ti = r1 / tci, tj = r1 - ti * tci;
                const half_t* ab = Ds + (g * 8 + (p >> 2)) * SD + ti * 16 + (p & 3) * 4;
                const half_t* bb = Xs + ((size_t)tap * kPix + g * 8 + (p >> 2)) * SX + tj * 16 + (p & 3) * 4;
#pragma unroll
                for (int ks = 0; ks < kPix / 32; ++ks) {
                    const half8_t av = tr_frag(ab + (ks * 32) * SD, ab + (ks * 32 + 4) * SD);
                    const half8_t bv = tr_frag(bb + (ks * 32) * SX, bb + (ks * 32 + 4) * SX);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int vt = wave + 4 * t;
        if (vt >= ntile) continue;
        const int tap = vt / ntile1, r1 = vt - tap * ntile1;
        const int ti = r1 / tci, tj = r1 - ti * tci;
        const int ci = tj * 16 + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = ti * 16 + g * 4 + r;
            if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + (size_t)tap * a.dw_tap + (size_t)co * a.dw_stride + ci, acc[t][r]);
        }
    }
}

template <int TPW>
int launch_taps(const WgArgs2& a, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_taps_kernel<TPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(wgrad)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_taps_kernel<TPW>), grid, dim3(256), lds, s, a);
    return 0;
}

template <int TI, int NJ>
int launch_tr(const WgArgs2& a, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tr_kernel<TI, NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(wgrad)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_tr_kernel<TI, NJ>), grid, dim3(256), lds, s, a);
    return 0;
}

}  // namespace

// one launch over a channel chunk (Cin <= 256) of X; dw points at the chunk's first input channel, rows are dw_stride floats apart,
// taps dw_tap floats apart
static int wgrad_launch(const half_t* x, int x_stride, const half_t* dy, int dy_stride, int M, int Cin_all, int cchunk, int Cout, float* dw, int dw_stride, long long dw_tap,
                        int gather, int Ho, int Wo, int Hs, int Ws, int tap0, int ntaps, hipStream_t s) {
    WgArgs2 b;
    const int Cin = cchunk < Cin_all ? cchunk : Cin_all, nchunk = (Cin_all + cchunk - 1) / cchunk;
    b.x = x; b.dy = dy; b.dw = dw; b.M = M; b.Cin = Cin; b.Cout = Cout; b.x_stride = x_stride; b.dy_stride = dy_stride;
    b.gather = gather; b.Ho = Ho; b.Wo = Wo; b.Hs = Hs; b.Ws = Ws; b.tap0 = tap0; b.dw_stride = dw_stride; b.dw_tap = dw_tap;
    b.ntaps = ntaps; b.cchunk = cchunk; b.Cin_all = Cin_all;
    const int tci = (Cin + 15) / 16, tco_all = (Cout + 15) / 16;            // Cin <= 256: tci <= 16
    static const bool no_taps = getenv("MAF_WGRAD_TAP_PER_WG") != nullptr;     // A/B: one tap per workgroup everywhere
    // (measured, n at batch 32: 8 -> 24 on 640^2 422 -> 246 us; 24 -> 48 on 320^2 190 -> 252 us — with two channel tiles the nine gathers
    //  of X outweigh the eight saved passes over dY: first stem conv only)
    if (ntaps == 9 && tci == 1 && tco_all * 9 <= 64 && !no_taps) {
        const int steps = maf_cdiv(M, kPix);
        int gx = steps < 1024 ? steps : 1024;
        b.chunk = maf_cdiv(steps, gx) * kPix;
        gx = maf_cdiv(M, b.chunk);
        const size_t lds = ((size_t)9 * kPix * (tci * 16 + 8) + (size_t)kPix * (tco_all * 16 + 8)) * 2;
        const int tpw = maf_cdiv(tci * tco_all * 9, 4);
        int rc = tpw <= 5 ? launch_taps<5>(b, dim3(gx), lds, s) : tpw <= 8 ? launch_taps<8>(b, dim3(gx), lds, s) : launch_taps<16>(b, dim3(gx), lds, s);
        if (rc) return rc;
        return maf_check_hip(hipGetLastError(), "conv wgrad launch");
    }
    b.WJ = tci >= 4 ? 4 : tci >= 2 ? 2 : 1;
    const int NJ = (tci + b.WJ - 1) / b.WJ, WI = 4 / b.WJ;                 // 1 .. 4
    const int want = (tco_all + WI - 1) / WI;
    const int TI = want >= 3 ? 4 : want;                                   // 1, 2, 4 (16 accumulator tiles per wave at most)
    const int gy = (tco_all + WI * TI - 1) / (WI * TI), gz = ntaps * nchunk;
    // Pixel chunks (grid x).  A chunk costs ~3 us per 64-pixel step it walks and, at its end, Cout * Cin atomics per tap at ~120 G/s
    // chip-wide: T(gx) ~ steps / gx * 3 us + gx * atomics / 120 G/s is smallest at gx = sqrt(steps * 3 us * 120 G/s / atomics); no more
    // workgroups than ~4 per CU (tools/wgrad_sweep.py: 20x20 768 -> 384 wants 32 chunks, 80x80 256 -> 128 wants 256, 160x160 72 -> 48 all it can get)
    const int steps = maf_cdiv(M, kPix);
    const double atomics = (double)(tco_all * 16) * (tci * 16);
    int gx = (int)std::sqrt((double)steps * 3.0 * 120e3 / atomics);
    const int fill = 1024 / (gy * gz) > 0 ? 1024 / (gy * gz) : 1;
    if (gx > fill) gx = fill;
    if (const char* e = getenv("MAF_WGRAD_GX")) gx = atoi(e);              // tools/wgrad_sweep.py
    if (gx > steps) gx = steps;
    if (gx < 1) gx = 1;
    b.chunk = maf_cdiv(steps, gx) * kPix;
    gx = maf_cdiv(M, b.chunk);
    const size_t lds = (size_t)kPix * (b.WJ * NJ * 16 + 8 + WI * TI * 16 + 8) * 2;
    const dim3 grid(gx, gy, gz);
    int rc = -1;
#define MAF_WG(T, N) if (TI == T && NJ == N) rc = launch_tr<T, N>(b, grid, lds, s);
    MAF_WG(1, 1) MAF_WG(1, 2) MAF_WG(1, 3) MAF_WG(1, 4) MAF_WG(2, 1) MAF_WG(2, 2) MAF_WG(2, 3) MAF_WG(2, 4) MAF_WG(4, 1) MAF_WG(4, 2) MAF_WG(4, 3) MAF_WG(4, 4)
#undef MAF_WG
    if (rc) return rc;
    return maf_check_hip(hipGetLastError(), "conv wgrad launch");
}

// dW (fp32, ACCUMULATED into: zero it first) of a conv with kernel k in {1, 3}, stride in {1, 2} (k = 3 needs stride 2, pad 1; k = 1 stride 2
// is pad 0): x [B,Hs,Ws,Cin] NHWC with pixel stride x_stride, dy [B,Ho,Wo,Cout] with pixel stride dy_stride.
// Layout of dW: k = 1 -> [Cout][Cin];  k = 3 -> TAP-MAJOR [3][3][Cout][Cin] (permute(2, 3, 0, 1) gives the framework's [Cout][Cin][3][3]).
extern "C" int maf_conv_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t B, int32_t Ho, int32_t Wo, int32_t Hs, int32_t Ws,
                              int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t dtype, float* dw, maf_stream_t stream) {
    MAF_REQUIRE(x && dy && dw && B > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0, "conv_wgrad: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16, "conv_wgrad: fp16 activations / gradients (fp32 parity mode runs the framework's GEMM)");
    MAF_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && x_stride % 8 == 0 && dy_stride % 8 == 0, "conv_wgrad: channels and strides must be multiples of 8");
    MAF_REQUIRE((ksize == 1 && (stride == 1 || stride == 2)) || (ksize == 3 && stride == 2), "conv_wgrad: k = 1 (stride 1 / 2) or k = 3 stride 2");
    if (stride == 1) MAF_REQUIRE(Hs == Ho && Ws == Wo, "conv_wgrad: stride 1 keeps the grid");
    else MAF_REQUIRE((Hs - 1) / 2 + 1 == Ho && (Ws - 1) / 2 + 1 == Wo, "conv_wgrad: Ho,Wo must equal floor((Hs-1)/2)+1");
    MAF_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv_wgrad: too many pixels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = B * Ho * Wo, kk = ksize * ksize;
    const int gather = stride == 2, tap0 = ksize == 1 ? 4 : 0;          // 1x1 stride 2 pad 0 reads (2 oy, 2 ox): the centre tap of the pad-1 geometry
    // the LDS tile holds <= 256 input channels: wider inputs as equal channel chunks (whole 16-channel tiles) of ONE grid
    const int nchunk = (Cin + 255) / 256, cchunk = ((Cin + nchunk - 1) / nchunk + 15) / 16 * 16;
    return wgrad_launch(static_cast<const half_t*>(x), x_stride, static_cast<const half_t*>(dy), dy_stride, M, Cin, cchunk, Cout,
                        dw, Cin, (long long)Cout * Cin, gather, Ho, Wo, Hs, Ws, tap0, kk, s);
}

extern "C" int maf_conv1x1_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t M, int32_t Cin,
                                 int32_t Cout, int32_t dtype, float* dw, maf_stream_t stream) {
    return maf_conv_wgrad(x, x_stride, dy, dy_stride, 1, 1, M, 1, M, Cin, Cout, 1, 1, dtype, dw, stream);
}
