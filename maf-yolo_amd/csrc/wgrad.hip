// Weight gradient of the 1x1 convolutions of the train-form graph:  dW[co][ci] = sum over pixels m of dY[m][co] * X[m][ci].
//
// Backward of nn.Conv2d(k=1) inside Conv / Head_DepthUni (yolov6/layers/common.py:29-50, 1331-1335) as the reference's
// autograd computes it in Trainer.train_in_steps (yolov6/core/engine.py:152-160).  As a GEMM it has a tiny output
// (Cout x Cin <= 576 x 576) and a reduction over up to 819 200 pixels; vendor TN GEMMs spend ~1 ms on it.  Here the
// pixel range is cut into chunks (one workgroup each, >= 1 per CU), a chunk is walked 64 pixels at a time: both NHWC tiles
// are transposed into LDS ([channel][pixel], so that an MFMA operand — 8 consecutive k = pixels of one channel — is ONE
// 16-byte LDS read), every wave owns a set of 16 x 16 output tiles whose accumulators stay in registers for the whole
// chunk, and the partial dW of the chunk is added to the fp32 result with atomics.  fp16 operands, fp32 accumulation.
//
// The same kernel is the weight gradient of the stride-2 convs (RepVGGBlock.rbr_dense / rbr_1x1 common.py:202-203, ConvWrapper :76-83):
// blockIdx.z walks the taps, and the X tile of tap (ky, kx) is gathered from pixel (2 oy - 1 + ky, 2 ox - 1 + kx) of the full-resolution
// input (zeros outside the image) while it is transposed into LDS — no im2col tensor, no per-tap copies.  Inputs wider than 256
// channels are cut into channel chunks by the host wrapper (the LDS tile holds one chunk).
#include <cstdlib>
#include "maf_common.h"

namespace {

struct WgArgs {
    const half_t* x; const half_t* dy; float* dw;
    int M, Cin, Cout, x_stride, dy_stride;
    int chunk;            // pixels per workgroup (multiple of 64)
    int co_blk;           // output-channel rows handled by one workgroup (blockIdx.y selects the block), multiple of 16
    // tap gather (gather != 0): pixel m = (b, oy, ox) on the Ho x Wo grid of dY reads X at (2 oy - 1 + ky, 2 ox - 1 + kx) of the Hs x Ws grid
    int gather, Ho, Wo, Hs, Ws, tap0;
    int dw_stride;        // row length of dW in (ci) elements
    int dw_es;            // element stride between consecutive ci of one tap (k*k), the tap index is added
    int tpw_taps;         // taps handled inside one workgroup (1, 3 or 9): dY is staged once per pixel step and reused for all of them
};

constexpr int kPix = 64;              // pixels per staging step
constexpr int kRow = kPix + 8;        // LDS row stride in halfs (144 B): rows start on different 16-byte bank slots

// TPW = 16 x 16 output tiles per wave
template <int TPW>
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int cinp = (a.Cin + 15) & ~15;
    const int co0 = blockIdx.y * a.co_blk;
    const int cob = min(a.co_blk, ((a.Cout - co0) + 15) & ~15);          // padded rows of this block
    half_t* Xs = reinterpret_cast<half_t*>(smem_raw);                    // [cinp][kRow]
    half_t* Ds = Xs + (size_t)cinp * kRow;                               // [cob][kRow]
    const int tci = cinp >> 4, tco = cob >> 4, ntile1 = tci * tco, ntile = ntile1 * a.tpw_taps;     // virtual tiles: (tap in workgroup, co tile, ci tile)

    f32x4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (f32x4_t)0.f;

    const int m_begin = blockIdx.x * a.chunk, m_end = min(a.M, m_begin + a.chunk);
    for (int m0 = m_begin; m0 < m_end; m0 += kPix) {
      for (int tt = 0; tt < a.tpw_taps; ++tt) {
        const int tap = a.tap0 + blockIdx.z * a.tpw_taps + tt, tky = tap / 3, tkx = tap - tky * 3;
        __syncthreads();                                                 // previous step's MFMAs have read the tiles
        // ---- stage: lane (g, p) of an item = pixel p of a 16-pixel group, channel chunk g of a 4-chunk group
        {
            const int xg = cinp >> 3;                                    // 8-channel chunks of X (incl. padding)
            for (int it = tid; it < (kPix / 16) * ((xg + 3) >> 2) * 64; it += 256) {
                const int l = it & 63, grp = it >> 6;
                const int pg = grp % (kPix / 16), cq = grp / (kPix / 16);
                const int px = pg * 16 + (l & 15), ch = (cq * 4 + (l >> 4)) * 8;
                if (ch < cinp) {
                    half8_t v = (half8_t)(half_t)0;
                    if (m0 + px < m_end && ch < a.Cin) {
                        if (!a.gather) {
                            v = *reinterpret_cast<const half8_t*>(a.x + (size_t)(m0 + px) * a.x_stride + ch);
                        } else {
                            const int m = m0 + px, ox = m % a.Wo, t2 = m / a.Wo, oy = t2 % a.Ho, bb = t2 / a.Ho;
                            const int iy = 2 * oy - 1 + tky, ix = 2 * ox - 1 + tkx;
                            if ((unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws)
                                v = *reinterpret_cast<const half8_t*>(a.x + ((size_t)(bb * a.Hs + iy) * a.Ws + ix) * a.x_stride + ch);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) Xs[(size_t)(ch + j) * kRow + px] = v[j];
                }
            }
            const int dg = tt == 0 ? cob >> 3 : 0;                        // dY: once per pixel step, shared by the taps of this workgroup
            for (int it = tid; it < (kPix / 16) * ((dg + 3) >> 2) * 64; it += 256) {
                const int l = it & 63, grp = it >> 6;
                const int pg = grp % (kPix / 16), cq = grp / (kPix / 16);
                const int px = pg * 16 + (l & 15), ch = (cq * 4 + (l >> 4)) * 8;
                if (ch < cob) {
                    half8_t v = (half8_t)(half_t)0;
                    if (m0 + px < m_end && co0 + ch < a.Cout) v = *reinterpret_cast<const half8_t*>(a.dy + (size_t)(m0 + px) * a.dy_stride + co0 + ch);
#pragma unroll
                    for (int j = 0; j < 8; ++j) Ds[(size_t)(ch + j) * kRow + px] = v[j];
                }
            }
        }
        __syncthreads();
        // ---- MFMA: D[co][ci] += sum_k dY^T[co][k] X^T... A = rows of Ds (co), B = rows of Xs (ci), k = pixel
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int vt = wave + 4 * t;
            if (vt < ntile && vt / ntile1 == tt) {
                const int tile = vt - tt * ntile1;
                const int ti = tile / tci, tj = tile - ti * tci;
#pragma unroll
                for (int ks = 0; ks < kPix / 32; ++ks) {
                    const half8_t av = *reinterpret_cast<const half8_t*>(Ds + (size_t)(ti * 16 + p) * kRow + ks * 32 + g * 8);
                    const half8_t bv = *reinterpret_cast<const half8_t*>(Xs + (size_t)(tj * 16 + p) * kRow + ks * 32 + g * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
      }
    }
    // ---- accumulator lane (g, p): rows co = 4g + r, column ci = p of its tile
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int vt = wave + 4 * t;
        if (vt >= ntile) continue;
        const int tt = vt / ntile1, tile = vt - tt * ntile1;
        const int tap = a.tap0 + blockIdx.z * a.tpw_taps + tt;
        const int ti = tile / tci, tj = tile - ti * tci;
        const int ci = tj * 16 + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + ti * 16 + g * 4 + r;
            if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + ((size_t)co * a.dw_stride + ci) * a.dw_es + (a.gather ? tap - a.tap0 : 0), acc[t][r]);
        }
    }
}

template <int T>
int launch_wg(const WgArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(wgrad)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((wgrad1x1_kernel<T>), grid, dim3(256), lds, s, a);
    return 0;
}

}  // namespace

static int wgrad_launch(const half_t* x, int x_stride, const half_t* dy, int dy_stride, int M, int Cin, int Cout, float* dw, int dw_stride, int dw_es,
                        int gather, int Ho, int Wo, int Hs, int Ws, int tap0, int ntaps, hipStream_t s) {
    WgArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.M = M; a.Cin = Cin; a.Cout = Cout; a.x_stride = x_stride; a.dy_stride = dy_stride;
    a.gather = gather; a.Ho = Ho; a.Wo = Wo; a.Hs = Hs; a.Ws = Ws; a.tap0 = tap0; a.dw_stride = dw_stride; a.dw_es = dw_es;
    const int cinp = (Cin + 15) & ~15, tci = cinp / 16;
    // rows of dW per workgroup: at most 16 tiles per wave (64 accumulator VGPRs) and 64 KiB of LDS together with X
    int tco = (64 / tci) > 0 ? (64 / tci) : 1;
    const int tco_all = (Cout + 15) / 16;
    if (tco > tco_all) tco = tco_all;
    while (tco > 1 && (size_t)(cinp + tco * 16) * kRow * 2 > 96 * 1024) --tco;
    MAF_REQUIRE((size_t)(cinp + tco * 16) * kRow * 2 <= 160 * 1024, "conv wgrad: Cin chunk too large for the LDS tile");
    a.co_blk = tco * 16;
    const int gy = maf_cdiv(tco_all, tco);
    // taps per workgroup: as many as the accumulator budget (16 tiles per wave) allows — dY is then read once for all of them
    // (measured on MI355X, n bs 32: looping the 9 taps inside a workgroup — dY staged once — is 20 % SLOWER than one tap per workgroup
    //  (5.1 vs 4.3 ms per step): the strided X gather dominates and nine times fewer workgroups hide its latency worse; kept selectable)
    a.tpw_taps = 1;
    if (ntaps == 9 && getenv("MAF_WGRAD_TAPS_IN_WG")) a.tpw_taps = tci * tco * 9 <= 64 ? 9 : tci * tco * 3 <= 64 ? 3 : 1;
    const int gz = ntaps / a.tpw_taps;
    int gx = 1024 / (gy * gz) > 0 ? 1024 / (gy * gz) : 1;              // ~4 workgroups per CU
    const int steps = maf_cdiv(M, kPix);
    if (gx > steps) gx = steps;
    a.chunk = maf_cdiv(steps, gx) * kPix;
    gx = maf_cdiv(M, a.chunk);
    const size_t lds = (size_t)(cinp + a.co_blk) * kRow * 2;
    const int tpw = maf_cdiv(tci * tco * a.tpw_taps, 4);
    const dim3 grid(gx, gy, gz);
    int rc;
    if (tpw <= 2) rc = launch_wg<2>(a, grid, lds, s);
    else if (tpw <= 4) rc = launch_wg<4>(a, grid, lds, s);
    else if (tpw <= 8) rc = launch_wg<8>(a, grid, lds, s);
    else rc = launch_wg<16>(a, grid, lds, s);
    if (rc) return rc;
    return maf_check_hip(hipGetLastError(), "conv wgrad launch");
}

// dW [Cout][Cin][k][k] (fp32, ACCUMULATED into: zero it first) of a conv with kernel k in {1, 3}, stride in {1, 2} (k = 3 needs stride 2, pad 1;
// k = 1 stride 2 is pad 0): x [B,Hs,Ws,Cin] NHWC with pixel stride x_stride, dy [B,Ho,Wo,Cout] with pixel stride dy_stride.
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
    for (int c0 = 0; c0 < Cin; c0 += 256) {                             // the LDS tile holds <= 256 input channels
        const int cc = std::min(256, Cin - c0);
        int rc = wgrad_launch(static_cast<const half_t*>(x) + c0, x_stride, static_cast<const half_t*>(dy), dy_stride, M, cc, Cout,
                              dw + (size_t)c0 * kk, Cin, kk, gather, Ho, Wo, Hs, Ws, tap0, kk, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int maf_conv1x1_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t M, int32_t Cin,
                                 int32_t Cout, int32_t dtype, float* dw, maf_stream_t stream) {
    return maf_conv_wgrad(x, x_stride, dy, dy_stride, 1, 1, M, 1, M, Cin, Cout, 1, 1, dtype, dw, stream);
}
