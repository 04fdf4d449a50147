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
    float* ws; int replicas; long long rsize;      // partial sums go to copy blockIdx.x % replicas of ws (rsize floats each, same layout as dw, zero on entry; folded into dw by wgrad3_fold_kernel); replicas = 1: ws = dw
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
    float* dwbase = a.ws + (size_t)(blockIdx.x % a.replicas) * a.rsize + zc * a.cchunk;
    const int Cin = min(a.cchunk, a.Cin_all - zc * a.cchunk);

    // staging items of this thread: (row, 8-channel chunk), the same for every step
    const int xg = cib >> 3, dg = cob >> 3;
    int xrow[XR], xch[XR], drow[DR], dch[DR], xdx[XR], xdy[XR];       // gather: (xdx, xdy) = the item's row as (columns, rows) of the output grid, added to the step's first pixel
#pragma unroll
    for (int u = 0; u < XR; ++u) {
        const int it = tid + 256 * u; xrow[u] = it / xg; xch[u] = (it - xrow[u] * xg) * 8;
        xdy[u] = a.gather ? xrow[u] / a.Wo : 0; xdx[u] = xrow[u] - xdy[u] * a.Wo;
        if (xrow[u] >= kPix) xrow[u] = -1;
    }
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
        // the step's first pixel as (image, row, column) of the output grid: ONE pair of divisions per step (m0 is uniform), not one per 16-byte item
        // (the 3 x 3 stride-2 layers spent more vector instructions on these quotients than on everything else)
        int ox0 = 0, oy0 = 0, bb0 = 0;
        if (a.gather) { const int t2 = m0 / a.Wo; ox0 = m0 - t2 * a.Wo; bb0 = t2 / a.Ho; oy0 = t2 - bb0 * a.Ho; }
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int m = m0 + xrow[u];
            if (xrow[u] >= 0 && m < m_end && xch[u] < Cin) {
                if (!a.gather) {
                    v = *reinterpret_cast<const half8_t*>(xbase + (size_t)m * a.x_stride + xch[u]);
                } else {
                    int ox = ox0 + xdx[u], oy = oy0 + xdy[u], bb = bb0;
                    if (ox >= a.Wo) { ox -= a.Wo; ++oy; }
                    while (oy >= a.Ho) { oy -= a.Ho; ++bb; }
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
    // staging items of this thread, the same for every step: (tap, row, channel chunk) and the row as (columns, rows) of the output grid
    int xch[XR], xrw[XR], xdx[XR], xdy[XR], xky[XR], xkx[XR], xso[XR], drw[DR], dch[DR];
#pragma unroll
    for (int u = 0; u < DR; ++u) { const int it = tid + 256 * u; drw[u] = it < nd ? it / dg : -1; dch[u] = (it % dg) * 8; }
#pragma unroll
    for (int u = 0; u < XR; ++u) {
        const int it = tid + 256 * u;
        const int r2 = it / xg, tap = r2 / kPix;
        xch[u] = (it - r2 * xg) * 8; xrw[u] = it < nx ? r2 - tap * kPix : -1;
        xso[u] = r2 * SX + xch[u];                                           // r2 = tap * kPix + row
        xky[u] = tap / 3; xkx[u] = tap - xky[u] * 3;
        xdy[u] = (r2 - tap * kPix) / a.Wo; xdx[u] = (r2 - tap * kPix) - xdy[u] * a.Wo;
    }
    auto fetch = [&](int m0) {
        const int t0 = m0 / a.Wo, ox0 = m0 - t0 * a.Wo, bb0 = t0 / a.Ho, oy0 = t0 - bb0 * a.Ho;      // one pair of divisions per step, not per item
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            if (xrw[u] >= 0) {
                const int ch = xch[u];
                const int m = m0 + xrw[u];
                if (m < m_end && ch < a.Cin) {
                    int ox = ox0 + xdx[u], oy = oy0 + xdy[u], bb = bb0;
                    if (ox >= a.Wo) { ox -= a.Wo; ++oy; }
                    while (oy >= a.Ho) { oy -= a.Ho; ++bb; }
                    const int iy = 2 * oy - 1 + xky[u], ix = 2 * ox - 1 + xkx[u];
                    if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws)
                        v = *reinterpret_cast<const half8_t*>(a.x + ((size_t)(bb * a.Hs + iy) * a.Ws + ix) * a.x_stride + ch);
                }
            }
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            half8_t v = (half8_t)(half_t)0;
            if (drw[u] >= 0) {
                const int m = m0 + drw[u];
                if (m < m_end && dch[u] < a.Cout) v = *reinterpret_cast<const half8_t*>(a.dy + (size_t)m * a.dy_stride + dch[u]);
            }
            dr[u] = v;
        }
    };
    fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += kPix) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < XR; ++u)
            if (xrw[u] >= 0) *reinterpret_cast<half8_t*>(Xs + xso[u]) = xr[u];
#pragma unroll
        for (int u = 0; u < DR; ++u)
            if (drw[u] >= 0) *reinterpret_cast<half8_t*>(Ds + drw[u] * SD + dch[u]) = dr[u];
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

// PATCH FORM of the 3x3 stride-2 weight gradient, for the narrow layers on the big maps (the stem: 8 -> 24 on 640^2, 24 -> 48 on 320^2, 48 -> 64 / 48 on 160^2).  One tap
// per workgroup re-reads dY nine times and gathers X per tap (1 GB of L2 traffic for 236 MB of tensors on 24 -> 48: 194 us); nine gathered tiles per step (above) read dY
// once but still pull every input pixel 2.25 times through the address path, 5.5 KB of new bytes per workgroup and step — too little in flight to cover the latency.
// Here a workgroup stages the INPUT PATCH of a TH x 16 tile of output pixels once ((2 TH + 1) x 33 pixels, 1.10 - 1.16x the tile's own input) next to the dY tile, and the
// nine taps read it in place: the transposing LDS read takes a row address per lane, so the X operand of tap (ky, kx) for output pixel (ty, tx) is simply the patch pixel
// (2 ty + ky, 2 tx + kx).  The (tap, ci tile) items are dealt to the four waves, each item multiplies with all co tiles (dY fragments loaded once per k-step and wave);
// the next tile's patch and dY are in registers while the instructions of this one run.
struct Wg3Args {
    const half_t* x; const half_t* dy; float* dw;
    int B, Ho, Wo, Hs, Ws, Cin, Cout, x_stride, dy_stride;
    int tilesX, tilesY, ntile, ntj, nti, SX, SD;
    float* ws; int replicas;        // partial sums go to copy blockIdx.x % replicas of ws ([replicas][9][Cout][Cin], zero on entry); wgrad3_fold_kernel adds the copies to dw
};

template <int NIT, int NTI, int KS, int MAXXI, int MAXDI, int OCC, int NW>
__global__ __launch_bounds__(64 * NW, OCC) void wgrad3_patch_kernel(const Wg3Args a) {
    constexpr int TW = 16, TH = 2 * KS, PIX = 32 * KS, PW = 2 * TW + 1, PH = 2 * TH + 1, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int SX = a.SX, SD = a.SD;
    half_t* Xs = reinterpret_cast<half_t*>(smem_raw);                    // [PH * PW][SX]   the input patch, NHWC pixels
    half_t* Ds = Xs + PH * PW * SX;                                      // [PIX][SD]       the dY tile, pixel k = ty * 16 + tx
    {
        half8_t* z = reinterpret_cast<half8_t*>(smem_raw);
        const int n8 = (PH * PW * SX + PIX * SD) >> 3;
        for (int i = tid; i < n8; i += NT) z[i] = (half8_t)(half_t)0;  // channel padding (to whole 16-column tiles) stays zero
    }
    // staging items of this thread: 16-byte chunks of the patch / of the dY tile
    const int ncx = a.Cin >> 3, ncd = a.Cout >> 3;
    int xm[MAXXI], dm[MAXDI];                                           // (row << 20 | column << 8 | chunk) of the item, -1: none
#pragma unroll
    for (int u = 0; u < MAXXI; ++u) {
        const int idx = tid + NT * u, pp = idx / ncx, cq = idx - pp * ncx, py = pp / PW, px = pp - py * PW;
        xm[u] = pp < PH * PW ? (py << 20) | (px << 8) | cq : -1;
    }
#pragma unroll
    for (int u = 0; u < MAXDI; ++u) {
        const int idx = tid + NT * u, k = idx / ncd, cq = idx - k * ncd;
        dm[u] = k < PIX ? ((k >> 4) << 20) | ((k & 15) << 8) | cq : -1;
    }
    half8_t xr[MAXXI], dr[MAXDI];
    auto fetch = [&](int tile) {
        const int b = tile / (a.tilesY * a.tilesX), t2 = tile - b * (a.tilesY * a.tilesX), tyi = t2 / a.tilesX, txi = t2 - tyi * a.tilesX;
        const int oy0 = tyi * TH, ox0 = txi * TW, iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
        const half_t* xb = a.x + ((long long)((long long)b * a.Hs + iy0) * a.Ws + ix0) * a.x_stride;
        const half_t* db = a.dy + ((long long)((long long)b * a.Ho + oy0) * a.Wo + ox0) * a.dy_stride;
#pragma unroll
        for (int u = 0; u < MAXXI; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int py = xm[u] >> 20, px = (xm[u] >> 8) & 0xfff, cq = xm[u] & 0xff;
            if (xm[u] >= 0 && (unsigned)(iy0 + py) < (unsigned)a.Hs && (unsigned)(ix0 + px) < (unsigned)a.Ws)
                v = *reinterpret_cast<const half8_t*>(xb + (py * a.Ws + px) * a.x_stride + cq * 8);
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < MAXDI; ++u) {
            half8_t v = (half8_t)(half_t)0;
            const int ty = dm[u] >> 20, tx = (dm[u] >> 8) & 0xfff, cq = dm[u] & 0xff;
            if (dm[u] >= 0 && oy0 + ty < a.Ho && ox0 + tx < a.Wo) v = *reinterpret_cast<const half8_t*>(db + (ty * a.Wo + tx) * a.dy_stride + cq * 8);
            dr[u] = v;
        }
    };
    // operand addresses.  A (dY^T): rows k = 32 ks + 8 g + (p >> 2) (+ 4), columns 4 (p & 3) of co tile i.  B (X): the patch pixel of output pixel k, moved by the tap.
    const half_t* abase = Ds + (g * 8 + (p >> 2)) * SD + (p & 3) * 4;
    int plo[KS], phi[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = ks * 32 + g * 8 + (p >> 2), k1 = k0 + 4;
        plo[ks] = ((2 * (k0 >> 4)) * PW + 2 * (k0 & 15)) * SX;
        phi[ks] = ((2 * (k1 >> 4)) * PW + 2 * (k1 & 15)) * SX;
    }
    const int nitems = 9 * a.ntj;
    int boff[NIT];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int item = wave + NW * n, tap = item / a.ntj, tj = item - tap * a.ntj, ky = tap / 3, kx = tap - ky * 3;
        boff[n] = (ky * PW + kx) * SX + tj * 16 + (p & 3) * 4;
    }
    f32x4_t acc[NIT][NTI];
#pragma unroll
    for (int n = 0; n < NIT; ++n)
#pragma unroll
        for (int i = 0; i < NTI; ++i) acc[n][i] = (f32x4_t)0.f;

    int tile = blockIdx.x;
    if (tile < a.ntile) fetch(tile);
    for (; tile < a.ntile; tile += gridDim.x) {
        __syncthreads();                                                 // the previous tile's instructions have read the LDS tiles (first pass: the clear)
#pragma unroll
        for (int u = 0; u < MAXXI; ++u)
            if (xm[u] >= 0) *reinterpret_cast<half8_t*>(Xs + ((xm[u] >> 20) * PW + ((xm[u] >> 8) & 0xfff)) * SX + (xm[u] & 0xff) * 8) = xr[u];
#pragma unroll
        for (int u = 0; u < MAXDI; ++u)
            if (dm[u] >= 0) *reinterpret_cast<half8_t*>(Ds + ((dm[u] >> 20) * 16 + ((dm[u] >> 8) & 0xfff)) * SD + (dm[u] & 0xff) * 8) = dr[u];
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntile) fetch(tile + gridDim.x);    // in flight during the instructions below
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8_t av[NTI];
#pragma unroll
            for (int i = 0; i < NTI; ++i)
                if (i < a.nti) av[i] = tr_frag(abase + (ks * 32) * SD + i * 16, abase + (ks * 32 + 4) * SD + i * 16);
#pragma unroll
            for (int n = 0; n < NIT; ++n) {
                if (wave + NW * n < nitems) {
                    const half8_t bv = tr_frag(Xs + plo[ks] + boff[n], Xs + phi[ks] + boff[n]);
#pragma unroll
                    for (int i = 0; i < NTI; ++i)
                        if (i < a.nti) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i], bv, acc[n][i], 0, 0, 0);
                }
            }
        }
    }
    // accumulator lane (g, p): rows co = 4 g + r, column ci = p of its tile; tap-major result [9][Cout][Cin]
    float* out = a.ws + (size_t)(blockIdx.x % a.replicas) * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int item = wave + NW * n;
        if (item < nitems) {
            const int tap = item / a.ntj, tj = item - tap * a.ntj, ci = tj * 16 + p;
#pragma unroll
            for (int i = 0; i < NTI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = i * 16 + g * 4 + r;
                    if (i < a.nti && co < a.Cout && ci < a.Cin) atomicAdd(out + ((size_t)tap * a.Cout + co) * a.Cin + ci, acc[n][i][r]);
                }
        }
    }
}

// dw += sum of the copies; the copies are left zero for the next launch on this stream
__global__ __launch_bounds__(256) void wgrad3_fold_kernel(float* __restrict__ ws, int replicas, int n, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int r = 0; r < replicas; ++r) { v += ws[(size_t)r * n + i]; ws[(size_t)r * n + i] = 0.f; }
    dw[i] += v;
}

// The workgroups of the patch form end with 9 * Cout * Cin atomics each; a thousand workgroups adding into the same few hundred cache lines serialise (24 -> 48: 8 M atomics,
// 66 us at the ~120 G/s of that pattern), so they add into kWsCopies copies (workgroup index modulo) kept by the library per (device, stream), folded by one small launch.
constexpr int kWsCopies = 8, kWsFloats = 9 * 64 * 48;            // kWsFloats: the largest dW of the patch form / of the many-chunk rule below
// (Round 5, measured and dropped — tools/conv_wgrad_bench.py: the same copies for the WIDE layers, 3 x 3 stride 2 128 -> 128 ... 192 -> 192 and the 1 x 1 layers of the
// 20 x 20 / 40 x 40 maps with a dW of 32 K ... 330 K floats and 30 - 56 pixel chunks, in a 64 MB workspace: +3 ... +5 us on every one of them, 2756 -> 2823 us over the
// dense weight gradients of a step.  Their atomics land on 0.1 - 1.3 MB of distinct lines and are not what bounds them; the fold's pass over eight copies is pure cost.)
struct WsSlot { int dev; hipStream_t s; float* p; };
static WsSlot g_ws[16];
static int g_nws = 0;

static float* patch_workspace(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < g_nws; ++i)
        if (g_ws[i].dev == dev && g_ws[i].s == s) return g_ws[i].p;
    if (g_nws == 16) return nullptr;
    float* p = nullptr;
    if (hipMalloc(&p, sizeof(float) * kWsCopies * kWsFloats) != hipSuccess) return nullptr;
    if (hipMemsetAsync(p, 0, sizeof(float) * kWsCopies * kWsFloats, s) != hipSuccess) return nullptr;
    g_ws[g_nws++] = {dev, s, p};
    return p;
}

template <int NIT, int NTI, int KS, int MAXXI, int MAXDI, int OCC, int NW>
int launch_patch(Wg3Args& a, int gx, hipStream_t s) {
    constexpr int TW = 16, TH = 2 * KS, PIX = 32 * KS, PW = 2 * TW + 1, PH = 2 * TH + 1;
    a.tilesX = maf_cdiv(a.Wo, TW); a.tilesY = maf_cdiv(a.Ho, TH); a.ntile = a.B * a.tilesY * a.tilesX;
    const size_t lds = ((size_t)PH * PW * a.SX + (size_t)PIX * a.SD) * 2;
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_patch_kernel<NIT, NTI, KS, MAXXI, MAXDI, OCC, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(wgrad3)");
        if (rc) return rc;
        attr = true;
    }
    if (const char* e = getenv("MAF_WGRAD3_GX")) gx = atoi(e);
    if (gx > a.ntile) gx = a.ntile;
    a.replicas = kWsCopies;
    if (const char* e = getenv("MAF_WGRAD3_R")) a.replicas = atoi(e);
    if (a.replicas > 1) {
        a.ws = patch_workspace(s);
        if (!a.ws) { maf_set_error("conv_wgrad: no workspace for the partial sums"); return MAF_E_HIP; }
    } else { a.replicas = 1; a.ws = a.dw; }
    hipLaunchKernelGGL((wgrad3_patch_kernel<NIT, NTI, KS, MAXXI, MAXDI, OCC, NW>), dim3(gx), dim3(64 * NW), lds, s, a);
    if (a.replicas > 1) {
        const int n = 9 * a.Cout * a.Cin;
        hipLaunchKernelGGL(wgrad3_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a.ws, a.replicas, n, a.dw);
    }
    return maf_check_hip(hipGetLastError(), "conv wgrad (patch) launch");
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
    b.ntaps = ntaps; b.cchunk = cchunk; b.Cin_all = Cin_all; b.ws = dw; b.replicas = 1; b.rsize = 0;
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
    // many pixel chunks adding into a small dW: copies (see patch_workspace), folded behind the launch
    const long long nout = (long long)ntaps * Cout * Cin_all;
    // (tools/conv_wgrad_bench.py, MAF_WGRAD_R = 1 / 8: 72 -> 48 on 160^2 82 -> 59 us, 72 -> 24 67 -> 34, the stem's 1x1 stride 2 97 -> 64; the 40 x 40 layers walk
    //  ~150 chunks and lose 1 - 3 us to the fold: copies only from 256 chunks on; 64 copies are slower than 8)
    static const int rep_max = getenv("MAF_WGRAD_R") ? atoi(getenv("MAF_WGRAD_R")) : kWsCopies;
    long long R = (long long)kWsCopies * kWsFloats / nout;
    if (R > rep_max) R = rep_max;
    if (gx < 256 || R < kWsCopies) R = 1;
    b.ws = dw; b.replicas = 1; b.rsize = nout;
    if (R >= 2 && dw_stride == Cin_all && dw_tap == (long long)Cout * Cin_all) {
        b.ws = patch_workspace(s);
        if (!b.ws) { maf_set_error("conv_wgrad: no workspace for the partial sums"); return MAF_E_HIP; }
        b.replicas = (int)R;
    }
    const size_t lds = (size_t)kPix * (b.WJ * NJ * 16 + 8 + WI * TI * 16 + 8) * 2;
    const dim3 grid(gx, gy, gz);
    int rc = -1;
#define MAF_WG(T, N) if (TI == T && NJ == N) rc = launch_tr<T, N>(b, grid, lds, s);
    MAF_WG(1, 1) MAF_WG(1, 2) MAF_WG(1, 3) MAF_WG(1, 4) MAF_WG(2, 1) MAF_WG(2, 2) MAF_WG(2, 3) MAF_WG(2, 4) MAF_WG(4, 1) MAF_WG(4, 2) MAF_WG(4, 3) MAF_WG(4, 4)
#undef MAF_WG
    if (rc) return rc;
    if (b.replicas > 1) hipLaunchKernelGGL(wgrad3_fold_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, b.ws, b.replicas, (int)nout, dw);
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
    static const int patch = getenv("MAF_WGRAD3_PATCH") ? atoi(getenv("MAF_WGRAD3_PATCH")) : -1;      // 0: never, 1: wherever it fits, default: the big maps
    if (ksize == 3 && Cin <= 48 && Cout <= 64 && patch != 0 && (patch == 1 || M >= 100000)) {
        Wg3Args b;
        b.x = static_cast<const half_t*>(x); b.dy = static_cast<const half_t*>(dy); b.dw = dw; b.B = B; b.Ho = Ho; b.Wo = Wo; b.Hs = Hs; b.Ws = Ws;
        b.Cin = Cin; b.Cout = Cout; b.x_stride = x_stride; b.dy_stride = dy_stride;
        b.ntj = (Cin + 15) / 16; b.nti = (Cout + 15) / 16; b.SX = b.ntj * 16 + 8; b.SD = b.nti * 16 + 8;
        // (items per wave, co tiles, k-steps per tile, patch / dY chunks per thread, workgroups per CU, waves): the wider layers on eight waves and fewer workgroups —
        // every workgroup ends with 9 * Cout * Cin atomics (tools/conv_wgrad_bench.py with MAF_WGRAD3_GX: 48 -> 64 on 160^2 105 us with 1024 workgroups, 52 with 256)
        if (b.ntj <= 1 && b.nti <= 2) return launch_patch<3, 2, 4, 3, 2, 4, 4>(b, 1024, s);      // 17 x 33 patch pixels x 1 chunk = 561 items / 256 threads; dY 128 x 3 / 256
        if (b.ntj <= 2 && b.nti <= 3) return launch_patch<3, 3, 2, 2, 1, 2, 8>(b, 448, s);       // 9 x 33 x <= 3 chunks = 891 / 512; 64 x 6 / 512
        return launch_patch<4, 4, 2, 4, 1, 2, 8>(b, 256, s);                                      // 9 x 33 x <= 6 = 1782 / 512; 64 x 8 / 512
    }
    // the LDS tile holds <= 256 input channels: wider inputs as equal channel chunks (whole 16-channel tiles) of ONE grid
    const int nchunk = (Cin + 255) / 256, cchunk = ((Cin + nchunk - 1) / nchunk + 15) / 16 * 16;
    return wgrad_launch(static_cast<const half_t*>(x), x_stride, static_cast<const half_t*>(dy), dy_stride, M, Cin, cchunk, Cout,
                        dw, Cin, (long long)Cout * Cin, gather, Ho, Wo, Hs, Ws, tap0, kk, s);
}

extern "C" int maf_conv1x1_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t M, int32_t Cin,
                                 int32_t Cout, int32_t dtype, float* dw, maf_stream_t stream) {
    return maf_conv_wgrad(x, x_stride, dy, dy_stride, 1, 1, M, 1, M, Cin, Cout, 1, 1, dtype, dw, stream);
}
