// Depth-wise weight gradient on the matrix cores, for the maps where the vector kernel (train_ops.hip: dw_wgrad_kernel) is bound by its halo
// tiles and its per-tile round trips:  dW[c][ky][kx] = sum_{b,y,x} dY[b,y,x,c] * X[b,y+ky-P,x+kx-P,c]  (zero padding).
//
// Backward of the depth-wise convs of DepthBottleneckUni / the re-parameterised branch sets (yolov6/layers/common.py:806-896) as the
// reference's autograd computes it (yolov6/core/engine.py:152-160).
//
// As a matrix product, per channel:  one 16x16x32 instruction takes an X row r (and the T - 1 rows after it), 32 output columns x and gives
//     D[i = (t, kx)][j] = sum_x X[r + t][x + kx - P] * dY[rho_j][x],     rho_j = r + P + (T - 1) - j,
// which is the contribution of that row pair to dW[ky = t - (T - 1) + j][kx]  (dropped where ky is outside 0 .. K - 1).  T = 16 / K rows
// of X share the instruction (K = 9: 1, 7: 2, 5: 3, 3: 5); 81 / 98 / 75 / 45 of its 256 products are used.  Operand A is a 16-byte LDS read of
// the PLANAR X row at a 2-byte-aligned address x + kx (gfx950 reads unaligned LDS vectors), operand B an aligned one of the planar dY row:
// two LDS reads per instruction, no vector arithmetic at all — the vector kernel needs 2 S + K - 1 reads per 8 S K multiply-adds and
// stages a (TH + K - 1) x (TW + K - 1) halo tile per 8 x 16 outputs (3x on a 20 x 20 map).
//
// Workgroup = one 16-byte channel group (8 channels, 2 per wave) x a share of the (image, row band) tiles; a band is TH rows over the FULL
// width (segments of 32 columns), so X has no horizontal halo and its K - 1 vertical halo rows exist only in dY.  The
// NHWC -> planar transposition happens in the LDS stores (eight 2-byte stores per 16-byte global load); the loads of the next stage are
// in registers while the instructions of the current one run.  Workgroups that share 128-byte lines (the 8 channel groups of a pixel's line)
// are consecutive slots of ONE XCD, so a line is fetched into one L2.
#include "maf_common.h"

namespace {

constexpr int kLd = 6;               // 16-byte loads a thread holds per staged operand

struct DwMfArgs {
    const half_t* x; const half_t* dy; float* dw;
    int x_stride, dy_stride, B, H, W, C, K;
    int TH, XR, DR, nbands, nsplit, replicas, ngroups;
};

struct __attribute__((packed, aligned(4))) half8_u { half8_t v; };      // 4-byte-aligned 16 bytes: two ds_read2_b32

// SEGS = segments of 32 output columns (W <= 32 SEGS).  LDS: rows of 8 channel lines, a line RS = 32 SEGS + 8 halfs (the image row behind P zeros; the
// tail zeros), a row RWS = 8 RS + 8 halfs: 4 RWS / 8 = 36 (mod 64) dwords, so the 16 dY rows an operand-B read touches start in 16 different bank quads.
template <int SEGS>
__global__ __launch_bounds__(256) void dw_wgrad_mfma_kernel(const DwMfArgs a) {
    constexpr int RS = 32 * SEGS + 8, RWS = 8 * RS + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // block -> (XCD, slot): the 8 channel groups of one 128-byte line on one XCD, dispatched next to each other
    const int bid = blockIdx.x, xcd = bid & 7, t8 = bid >> 3, cgl = t8 & 7, u = (t8 >> 3) * 8 + xcd;
    const int nline = (a.ngroups + 7) >> 3;
    const int line = u % nline, split = u / nline;
    const int cg = line * 8 + cgl;
    if (split >= a.nsplit || cg >= a.ngroups) return;
    const int c0 = cg * 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int XCP = a.XR * RWS;
    half_t* Xp = reinterpret_cast<half_t*>(smem_raw);                    // [2][XR][8][RS]   copy 0: Xp[row][ch][P + x] = X[r0 + row][x][ch]; copy 1: the same one element to the left
    half_t* Dp = Xp + 2 * XCP;                                           // [DR][8][RS]
    {
        half8_t* z = reinterpret_cast<half8_t*>(smem_raw);
        const int n8 = (2 * a.XR + a.DR) * (RWS / 8);
        for (int i = tid; i < n8; i += 256) z[i] = (half8_t)(half_t)0;
    }
    const int K = a.K, P = K >> 1, T = 16 / K, THp = (a.TH + T - 1) / T * T, JV = K + T - 1, DRows = THp + K - 1;
    // staging items of this thread: cell idx = tid + 256 q -> (row, x) of a [rows][W] region; soff = its LDS offset (halfs) in a row-major [row][8][RS] block
    int srow[kLd], sx[kLd], soff[kLd];
#pragma unroll
    for (int q = 0; q < kLd; ++q) { const int idx = tid + 256 * q; srow[q] = idx / a.W; sx[q] = idx - srow[q] * a.W; soff[q] = srow[q] * RWS + sx[q]; }

    f32x4_t acc0 = (f32x4_t)0.f, acc1 = (f32x4_t)0.f;
    const int ntile = a.B * a.nbands;
    half8_t xr[kLd], dr[kLd];
    // rows [row0, row0 + nrows) of image bi (zero outside the image) of src, this workgroup's channel group
    auto fetch = [&](half8_t* regs, const half_t* src, int stride, int bi, int row0, int nrows) {
#pragma unroll
        for (int q = 0; q < kLd; ++q) {
            half8_t v = (half8_t)(half_t)0;
            const int iy = row0 + srow[q];
            if (srow[q] < nrows && (unsigned)iy < (unsigned)a.H)
                v = *reinterpret_cast<const half8_t*>(src + ((size_t)((size_t)bi * a.H + iy) * a.W + sx[q]) * stride + c0);
            regs[q] = v;
        }
    };
    const int ii = min(p, T * K - 1), ti = ii / K, kxi = ii - ti * K, jj = min(p, JV - 1);
    const half_t* ap = Xp + (kxi & 1) * XCP + ti * RWS + (wave * 2) * RS + (kxi & ~1) + 8 * g;      // odd shifts: the copy one element to the left, at the even offset below
    const half_t* bp = Dp + (T - 1 - jj + K - 1) * RWS + (wave * 2) * RS + 8 * g;
    const int nr = THp / T;

    int tile = split;
    if (tile < ntile) {
        const int bi = tile / a.nbands, r0 = (tile - bi * a.nbands) * a.TH;
        fetch(xr, a.x, a.x_stride, bi, r0, a.TH);
        fetch(dr, a.dy, a.dy_stride, bi, r0 + P - (K - 1), DRows);
    }
    for (; tile < ntile; tile += a.nsplit) {
        const int nxt = tile + a.nsplit;
        __syncthreads();                                                 // the previous tile's instructions have read Xp / Dp (first pass: the clear)
        {
#pragma unroll
            for (int q = 0; q < kLd; ++q) {
                if (srow[q] < a.TH) {
                    half_t* d = Xp + soff[q] + P;
#pragma unroll
                    for (int c = 0; c < 8; ++c) d[c * RS] = xr[q][c];
                    half_t* d2 = d + XCP - 1;
#pragma unroll
                    for (int c = 0; c < 8; ++c) d2[c * RS] = xr[q][c];
                }
                if (srow[q] < DRows) {
                    half_t* d = Dp + soff[q];
#pragma unroll
                    for (int c = 0; c < 8; ++c) d[c * RS] = dr[q][c];
                }
            }
        }
        __syncthreads();
        if (nxt < ntile) {                                               // the next tile's loads, in flight during the instructions below
            const int nbi = nxt / a.nbands, nr0 = (nxt - nbi * a.nbands) * a.TH;
            fetch(xr, a.x, a.x_stride, nbi, nr0, a.TH);
            fetch(dr, a.dy, a.dy_stride, nbi, nr0 + P - (K - 1), DRows);
        }
        {
#pragma unroll
            for (int s = 0; s < SEGS; ++s) {
                const half_t* as = ap + 32 * s;
                const half_t* bs = bp + 32 * s;
                for (int rr = 0; rr < nr; ++rr) {
                    const half8_t a0 = reinterpret_cast<const half8_u*>(as)->v, a1 = reinterpret_cast<const half8_u*>(as + RS)->v;
                    const half8_t b0 = *reinterpret_cast<const half8_t*>(bs), b1 = *reinterpret_cast<const half8_t*>(bs + RS);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc1, 0, 0, 0);
                    as += T * RWS;
                    bs += T * RWS;
                }
            }
        }
    }
    // accumulator lane (g, p): rows i = 4 g + r = (t, kx), column j = p  ->  dW[ky = t - (T - 1) + j][kx]
    float* dwr = a.dw + (size_t)(split % a.replicas) * a.C * (K * K) + (size_t)(c0 + wave * 2) * (K * K);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r, t = i / K, kx = i - t * K, ky = t - (T - 1) + p;
        if (i < T * K && p < JV && ky >= 0 && ky < K) {
            atomicAdd(dwr + ky * K + kx, acc0[r]);
            atomicAdd(dwr + K * K + ky * K + kx, acc1[r]);
        }
    }
}

template <int SEGS>
int launch_mf(const DwMfArgs& a, int blocks, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad_mfma_kernel<SEGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024), "hipFuncSetAttribute(dw_wgrad_mfma)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((dw_wgrad_mfma_kernel<SEGS>), dim3(blocks), dim3(256), lds, s, a);
    return 0;
}

}  // namespace

// k in {3, 5, 7, 9}, fp16, W <= 96; returns MAF_E_UNSUPPORTED (nothing launched, no error text) where the shape does not fit the kernel
int maf_dw_wgrad_mfma(const void* x, int x_stride, const void* dy, int dy_stride, int B, int H, int W, int C, int k, float* dw, int replicas, hipStream_t s) {
    if ((k != 3 && k != 5 && k != 7 && k != 9) || C % 8 || x_stride % 8 || dy_stride % 8 || W > 96) return MAF_E_UNSUPPORTED;
    DwMfArgs a;
    a.x = static_cast<const half_t*>(x); a.dy = static_cast<const half_t*>(dy); a.dw = dw; a.x_stride = x_stride; a.dy_stride = dy_stride;
    a.B = B; a.H = H; a.W = W; a.C = C; a.K = k; a.replicas = replicas;
    const int T = 16 / k, segs = maf_cdiv(W, 32), rws = 8 * (32 * segs + 8) + 8;
    a.ngroups = C / 8;
    // the band: as many rows as 64 KB of LDS (two workgroups per CU and more) and kLd loads per thread and operand allow
    int th = H;
    for (;; --th) {
        if (th < 1) return MAF_E_UNSUPPORTED;
        const int thp = (th + T - 1) / T * T, dr = thp + k - 1;
        if (2l * (2 * thp + dr) * rws <= 64 * 1024 && dr * W <= 256 * kLd) break;
    }
    a.nbands = maf_cdiv(H, th);
    a.TH = maf_cdiv(H, a.nbands);                                           // balanced bands
    const int thp = (a.TH + T - 1) / T * T;
    a.XR = thp;
    a.DR = thp + k - 1;
    const size_t lds = 2 * (size_t)(2 * a.XR + a.DR) * rws;
    const int ntile = B * a.nbands;
    int want = 768;
    if (const char* e = getenv("MAF_DWMF_WG")) want = atoi(e);
    int nsplit = want / a.ngroups;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntile) nsplit = ntile;
    a.nsplit = nsplit;
    const int nline = (a.ngroups + 7) / 8, units = nline * nsplit, blocks = maf_cdiv(units, 8) * 64;
    const int rc = segs == 1 ? launch_mf<1>(a, blocks, lds, s) : segs == 2 ? launch_mf<2>(a, blocks, lds, s) : launch_mf<3>(a, blocks, lds, s);
    if (rc) return rc;
    return maf_check_hip(hipGetLastError(), "dw_wgrad_mfma launch");
}
