// Weight gradient of the 1x1 convolutions of the train-form graph:  dW[co][ci] = sum over pixels m of dY[m][co] * X[m][ci].
//
// Backward of nn.Conv2d(k=1) inside Conv / Head_DepthUni (yolov6/layers/common.py:29-50, 1331-1335) as the reference's
// autograd computes it in Trainer.train_in_steps (yolov6/core/engine.py:152-160).  As a GEMM it has a tiny output
// (Cout x Cin <= 576 x 576) and a reduction over up to 819 200 pixels; vendor TN GEMMs spend ~1 ms on it.  Here the
// pixel range is cut into chunks (one workgroup each, >= 1 per CU), a chunk is walked 64 pixels at a time: both NHWC tiles
// are transposed into LDS ([channel][pixel], so that an MFMA operand — 8 consecutive k = pixels of one channel — is ONE
// 16-byte LDS read), every wave owns a set of 16 x 16 output tiles whose accumulators stay in registers for the whole
// chunk, and the partial dW of the chunk is added to the fp32 result with atomics.  fp16 operands, fp32 accumulation.
#include "maf_common.h"

namespace {

struct WgArgs {
    const half_t* x; const half_t* dy; float* dw;
    int M, Cin, Cout, x_stride, dy_stride;
    int chunk;            // pixels per workgroup (multiple of 64)
    int co_blk;           // output-channel rows handled by one workgroup (blockIdx.y selects the block), multiple of 16
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
    const int tci = cinp >> 4, tco = cob >> 4, ntile = tci * tco;

    f32x4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (f32x4_t)0.f;

    const int m_begin = blockIdx.x * a.chunk, m_end = min(a.M, m_begin + a.chunk);
    for (int m0 = m_begin; m0 < m_end; m0 += kPix) {
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
                    if (m0 + px < m_end && ch < a.Cin) v = *reinterpret_cast<const half8_t*>(a.x + (size_t)(m0 + px) * a.x_stride + ch);
#pragma unroll
                    for (int j = 0; j < 8; ++j) Xs[(size_t)(ch + j) * kRow + px] = v[j];
                }
            }
            const int dg = cob >> 3;
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
            const int tile = wave + 4 * t;
            if (tile < ntile) {
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
    // ---- accumulator lane (g, p): rows co = 4g + r, column ci = p of its tile
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile >= ntile) continue;
        const int ti = tile / tci, tj = tile - ti * tci;
        const int ci = tj * 16 + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + ti * 16 + g * 4 + r;
            if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + (size_t)co * a.Cin + ci, acc[t][r]);
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

extern "C" int maf_conv1x1_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t M, int32_t Cin,
                                 int32_t Cout, int32_t dtype, float* dw, maf_stream_t stream) {
    MAF_REQUIRE(x && dy && dw && M > 0 && Cin > 0 && Cout > 0, "conv1x1_wgrad: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16, "conv1x1_wgrad: fp16 activations / gradients (fp32 runs the GEMM of the framework)");
    MAF_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && x_stride % 8 == 0 && dy_stride % 8 == 0, "conv1x1_wgrad: channels and strides must be multiples of 8");
    WgArgs a;
    a.x = static_cast<const half_t*>(x); a.dy = static_cast<const half_t*>(dy); a.dw = dw;
    a.M = M; a.Cin = Cin; a.Cout = Cout; a.x_stride = x_stride; a.dy_stride = dy_stride;
    const int cinp = (Cin + 15) & ~15, tci = cinp / 16;
    // rows of dW per workgroup: at most 16 tiles per wave (64 accumulator VGPRs) and 64 KiB of LDS together with X
    int tco = (64 / tci) > 0 ? (64 / tci) : 1;
    const int tco_all = (Cout + 15) / 16;
    if (tco > tco_all) tco = tco_all;
    while (tco > 1 && (size_t)(cinp + tco * 16) * kRow * 2 > 96 * 1024) --tco;
    MAF_REQUIRE((size_t)(cinp + tco * 16) * kRow * 2 <= 160 * 1024, "conv1x1_wgrad: Cin too large for the LDS tile");
    a.co_blk = tco * 16;
    const int gy = maf_cdiv(tco_all, tco);
    int gx = 1024 / gy > 0 ? 1024 / gy : 1;                             // ~4 workgroups per CU
    const int steps = maf_cdiv(M, kPix);
    if (gx > steps) gx = steps;
    a.chunk = maf_cdiv(steps, gx) * kPix;
    gx = maf_cdiv(M, a.chunk);
    const size_t lds = (size_t)(cinp + a.co_blk) * kRow * 2;
    const int tpw = maf_cdiv(tci * tco, 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(gx, gy);
    int rc;
    if (tpw <= 2) rc = launch_wg<2>(a, grid, lds, s);
    else if (tpw <= 4) rc = launch_wg<4>(a, grid, lds, s);
    else if (tpw <= 8) rc = launch_wg<8>(a, grid, lds, s);
    else rc = launch_wg<16>(a, grid, lds, s);
    if (rc) return rc;
    return maf_check_hip(hipGetLastError(), "conv1x1_wgrad launch");
}
