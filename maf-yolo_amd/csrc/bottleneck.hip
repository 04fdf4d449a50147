// Fused DepthBottleneckUni (deploy form): 1x1 (c -> 3c) + SiLU -> depth-wise k x k + SiLU -> 1x1 (3c -> c) + SiLU in ONE
// kernel, the two 3c-channel intermediates never leave the CU.
//
// Replaces, for one bottleneck, Conv.forward_fuse x2 + the merged UniRepLKNetBlock + DepthBottleneckUni.act of
// yolov6/layers/common.py:918-927 (conv1 :905, conv2 :908, act :909, one_conv :910) — three launches that wrote and
// re-read the 3c-wide tensors twice (4.9 MB x 2 per image at 80x80: SURVEY.md §8 a5 "prime fusion target").
//
// One workgroup = one TH x TW output tile of one image.  LDS holds the c-channel input halo tile X, and per block of
// 64 mid channels the 1x1 output T1 on the halo tile and the depth-wise output T2 on the tile:
//   for each mid block:   A. T1 = SiLU(X * W1[:, block] + b1)      MFMA, A fragments = ds_read_b128 of X rows, zero outside the image
//                         B. T2 = SiLU(DW_k(T1) + bdw)             VALU (v_fma_mix_f32), 4-pixel strips per lane as in dwconv.hip
//                         C. acc += T2 * W2[block, :]              MFMA, accumulators stay in registers across the blocks
//   epilogue: out = SiLU(acc + b2), contiguous NHWC rows into the concat slice.
// The 1x1 on the halo is recomputed ((TH+k-1)(TW+k-1)/(TH*TW) times) — cheap, its K is only c — in exchange for never
// writing T1/T2: per bottleneck the HBM traffic drops from (2c + 12c) to 2c channels per pixel.
// fp16 storage, fp32 accumulation everywhere (same arithmetic as the three separate kernels up to the fp16 rounding of
// T1/T2, which those kernels apply as well when they store them).
#include "maf_common.h"

namespace {

constexpr int MB = 64;                 // mid channels per block
constexpr int PADH = 8;                // LDS row pad in halfs (16 B): consecutive rows start on different bank groups

struct BnArgs {
    const half_t* x; half_t* out;
    const half8_t* w1; const half8_t* w2; const half_t* wdw;
    const float* b1; const float* bdw; const float* b2;
    int B, H, W, Cin, Cout, nMB, x_stride, x_coff, out_stride, out_coff;
    int TH, TW, tilesX, tilesY, nwg, act_mid;
};

__device__ __forceinline__ void vmac8(float (&acc)[8], const half8_t& v, const half8_t& w) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, v), b = __builtin_bit_cast(u32x4_t, w);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q]) : "v"(a[q]), "v"(b[q]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q + 1]) : "v"(a[q]), "v"(b[q]));
    }
}

template <int K, int CT2, int MT2>     // MT2 = output m-tiles (16 pixels) per wave = TH*TW/64
__global__ __launch_bounds__(256) void bottleneck_kernel(const BnArgs a) {
    constexpr int P = K / 2, R = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tx = lid % a.tilesX;
    int t = lid / a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    const int RH = a.TH + K - 1, RW = a.TW + K - 1, NP = RH * RW, NPT = (NP + 15) >> 4;
    const int steps1 = (a.Cin + 31) >> 5;
    const int XS = steps1 * 32 + PADH, TS = MB + PADH;            // LDS row strides (halfs); X rows padded to whole k-steps
    half_t* Xs = reinterpret_cast<half_t*>(smem_raw);             // [NP][XS]
    half_t* T1 = Xs + (size_t)NP * XS;                            // [NP][TS]
    half_t* T2 = T1 + (size_t)NP * TS;                            // [TH*TW][TS]
    half_t* Wd = T2 + (size_t)a.TH * a.TW * TS;                   // [K*K][MB]

    {   // ---- stage the input halo tile (zero outside the image and beyond Cin up to whole k-steps)
        const int cgs = steps1 * 4;                                // 16-byte chunks per row incl. k-step padding
        const half_t* xin = a.x + a.x_coff;
        for (int idx = tid; idx < NP * cgs; idx += 256) {
            const int cg = idx % cgs, pix = idx / cgs;
            const int rx = pix % RW, ry = pix / RW;
            const int iy = y0 - P + ry, ix = x0 - P + rx;
            half8_t v = (half8_t)(half_t)0;
            if (cg * 8 < a.Cin && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const half8_t*>(xin + ((size_t)((size_t)b * a.H + iy) * a.W + ix) * a.x_stride + cg * 8);
            *reinterpret_cast<half8_t*>(Xs + (size_t)pix * XS + cg * 8) = v;
        }
    }
    f32x4_t acc2[MT2][CT2];
#pragma unroll
    for (int i = 0; i < MT2; ++i)
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) acc2[i][ct] = (f32x4_t)0.f;
    __syncthreads();

    for (int mb = 0; mb < a.nMB; ++mb) {
        // ---- A. T1 = act(X * W1[:, block] + b1) on the halo tile; exact zeros outside the image (the DW's padding)
        for (int i = tid; i < K * K * (MB / 8); i += 256)
            *reinterpret_cast<half8_t*>(Wd + i * 8) = *reinterpret_cast<const half8_t*>(a.wdw + (size_t)(i / (MB / 8)) * (a.nMB * MB) + mb * MB + (i % (MB / 8)) * 8);
        const float bl0 = a.b1[mb * MB + p * 4 + 0], bl1 = a.b1[mb * MB + p * 4 + 1], bl2 = a.b1[mb * MB + p * 4 + 2], bl3 = a.b1[mb * MB + p * 4 + 3];
        for (int mt = wave; mt < NPT; mt += 4) {
            f32x4_t acc1[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc1[ct] = (f32x4_t)0.f;
            const int prow = min(mt * 16 + p, NP - 1);
            for (int ks = 0; ks < steps1; ++ks) {
                const half8_t av = *reinterpret_cast<const half8_t*>(Xs + (size_t)prow * XS + ks * 32 + g * 8);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const half8_t bv = a.w1[((size_t)(mb * 4 + ct) * steps1 + ks) * 64 + lane];
                    acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc1[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pix = mt * 16 + g * 4 + r;
                if (pix < NP) {
                    const int rx = pix % RW, ry = pix / RW;
                    const bool in = (unsigned)(y0 - P + ry) < (unsigned)a.H && (unsigned)(x0 - P + rx) < (unsigned)a.W;
                    half4_t h;
                    h[0] = in ? (half_t)maf_act<MAF_ACT_SILU>(acc1[0][r] + bl0) : (half_t)0;
                    h[1] = in ? (half_t)maf_act<MAF_ACT_SILU>(acc1[1][r] + bl1) : (half_t)0;
                    h[2] = in ? (half_t)maf_act<MAF_ACT_SILU>(acc1[2][r] + bl2) : (half_t)0;
                    h[3] = in ? (half_t)maf_act<MAF_ACT_SILU>(acc1[3][r] + bl3) : (half_t)0;
                    *reinterpret_cast<half4_t*>(T1 + (size_t)pix * TS + p * 4) = h;
                }
            }
        }
        __syncthreads();
        // ---- B. T2 = act(DW_k(T1) + bdw): lane = one 8-channel group x a 4-pixel strip of one tile row
        {
            const int NSX = a.TW / R;
            const int items = a.TH * NSX * (MB / 8);
            for (int it = tid; it < items; it += 256) {
                const int cgi = it % (MB / 8), u = it / (MB / 8);
                const int s = u % NSX, ry = u / NSX;
                float acc[R][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float bv = a.bdw[mb * MB + cgi * 8 + j];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][j] = bv;
                }
#pragma unroll 1
                for (int ky = 0; ky < K; ++ky) {
                    const half_t* row = T1 + (size_t)((ry + ky) * RW + s * R) * TS + cgi * 8;
                    half8_t wv[K];
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) wv[kx] = *reinterpret_cast<const half8_t*>(Wd + (ky * K + kx) * MB + cgi * 8);
#pragma unroll
                    for (int i = 0; i < R + K - 1; ++i) {
                        const half8_t v = *reinterpret_cast<const half8_t*>(row + (size_t)i * TS);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int kx = i - r;
                            if (kx >= 0 && kx < K) vmac8(acc[r], v, wv[kx]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    half8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)(a.act_mid ? maf_act<MAF_ACT_SILU>(acc[r][j]) : acc[r][j]);
                    *reinterpret_cast<half8_t*>(T2 + (size_t)(ry * a.TW + s * R + r) * TS + cgi * 8) = o;
                }
            }
        }
        __syncthreads();
        // ---- C. acc2 += T2 * W2[block, :]
#pragma unroll
        for (int i = 0; i < MT2; ++i) {
            const int mt2 = wave * MT2 + i;
#pragma unroll
            for (int ks = 0; ks < MB / 32; ++ks) {
                const half8_t av = *reinterpret_cast<const half8_t*>(T2 + (size_t)(mt2 * 16 + p) * TS + ks * 32 + g * 8);
#pragma unroll
                for (int ct = 0; ct < CT2; ++ct) {
                    const half8_t bv = a.w2[((size_t)(mb * CT2 + ct) * (MB / 32) + ks) * 64 + lane];
                    acc2[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc2[i][ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: out = SiLU(acc2 + b2); lane (g, p) owns channels p*CT2 .. p*CT2+CT2-1 of 4 pixels per m-tile
    float bias2[CT2];
#pragma unroll
    for (int ct = 0; ct < CT2; ++ct) bias2[ct] = a.b2[p * CT2 + ct];
    const int nvalid = a.Cout - p * CT2;
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = (wave * MT2 + i) * 16 + g * 4 + r;
            const int oy = y0 + idx / a.TW, ox = x0 + idx % a.TW;
            if (oy >= a.H || ox >= a.W) continue;
            half_t* op = a.out + ((size_t)((size_t)b * a.H + oy) * a.W + ox) * a.out_stride + a.out_coff + p * CT2;
            if (nvalid >= CT2) {
                uint32_t w[CT2 / 2];
#pragma unroll
                for (int q = 0; q < CT2 / 2; ++q) {
                    const half2_t h = {(half_t)maf_act<MAF_ACT_SILU>(acc2[i][2 * q][r] + bias2[2 * q]), (half_t)maf_act<MAF_ACT_SILU>(acc2[i][2 * q + 1][r] + bias2[2 * q + 1])};
                    w[q] = __builtin_bit_cast(uint32_t, h);
                }
                if (CT2 == 4) *reinterpret_cast<u32x2_t*>(op) = (u32x2_t){w[0], w[1 % (CT2 / 2)]};
                else *reinterpret_cast<uint32_t*>(op) = w[0];
            } else {
#pragma unroll
                for (int ct = 0; ct < CT2; ++ct)
                    if (ct < nvalid) op[ct] = (half_t)maf_act<MAF_ACT_SILU>(acc2[i][ct][r] + bias2[ct]);
            }
        }
    }
}

size_t bn_lds(int TH, int TW, int K, int Cin) {
    const size_t NP = (size_t)(TH + K - 1) * (TW + K - 1);
    const int steps1 = (Cin + 31) / 32;
    return (NP * (size_t)(steps1 * 32 + PADH) + NP * (MB + PADH) + (size_t)TH * TW * (MB + PADH) + (size_t)K * K * MB) * 2;
}

template <int K>
int launch_k(const BnArgs& a, size_t lds, hipStream_t s) {
    const int ct2 = a.Cout <= 32 ? 2 : 4;
    const int mt2 = a.TH * a.TW / 64;
#define MAF_BN(C2, M2)                                                                                                                   \
    if (ct2 == C2 && mt2 == M2) {                                                                                                        \
        static bool attr = false;                                                                                                        \
        if (!attr) {                                                                                                                     \
            int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_kernel<K, C2, M2>),                     \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(bottleneck)"); \
            if (rc) return rc;                                                                                                           \
            attr = true;                                                                                                                 \
        }                                                                                                                                \
        hipLaunchKernelGGL((bottleneck_kernel<K, C2, M2>), dim3(a.nwg), dim3(256), lds, s, a);                                           \
        return maf_check_hip(hipGetLastError(), "bottleneck launch");                                                                    \
    }
    MAF_BN(2, 1) MAF_BN(2, 2) MAF_BN(2, 4) MAF_BN(4, 1) MAF_BN(4, 2) MAF_BN(4, 4)
#undef MAF_BN
    maf_set_error("bottleneck: unsupported tile (TH*TW must be 64, 128 or 256)");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

int maf_launch_bottleneck(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "bottleneck: fp16 only (the fp32 parity mode runs the three kernels separately)");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && op->out, "bottleneck: one direct source");
    MAF_REQUIRE(op->Cin % 8 == 0 && op->Cin <= 64 && op->Cout % 2 == 0 && op->Cout <= 64, "bottleneck: c <= 64 channels in and out");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 4 == 0 && op->out_coff % 4 == 0, "bottleneck: stride/offset alignment");
    MAF_REQUIRE(op->tile_k > 0 && op->w && op->bias && op->aux[0] && op->aux[1] && op->aux[2] && op->aux[3], "bottleneck: null weights (w=W1, bias=b1, aux = {wdw, bdw, W2, b2}), tile_k = mid blocks");
    BnArgs a;
    a.x = static_cast<const half_t*>(sr.ptr); a.out = static_cast<half_t*>(op->out);
    a.w1 = static_cast<const half8_t*>(op->w); a.b1 = op->bias;
    a.wdw = static_cast<const half_t*>(op->aux[0]); a.bdw = static_cast<const float*>(op->aux[1]);
    a.w2 = static_cast<const half8_t*>(op->aux[2]); a.b2 = static_cast<const float*>(op->aux[3]);
    a.B = op->B; a.H = op->H; a.W = op->W; a.Cin = op->Cin; a.Cout = op->Cout; a.nMB = op->tile_k;
    a.x_stride = sr.stride; a.x_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.act_mid = op->act == MAF_ACT_SILU;
    a.TH = op->tile_p > 0 ? op->tile_p : 8; a.TW = op->tile_c > 0 ? op->tile_c : 16;
    MAF_REQUIRE(a.TW % 4 == 0 && (a.TH * a.TW) % 64 == 0, "bottleneck: TW multiple of 4, TH*TW multiple of 64");
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH);
    a.nwg = a.B * a.tilesX * a.tilesY;
    const size_t lds = bn_lds(a.TH, a.TW, op->ksize, a.Cin);
    MAF_REQUIRE(lds <= 160 * 1024, "bottleneck: tile does not fit LDS");
    switch (op->ksize) {
        case 3: return launch_k<3>(a, lds, s);
        case 5: return launch_k<5>(a, lds, s);
        case 7: return launch_k<7>(a, lds, s);
        case 9: return launch_k<9>(a, lds, s);
        default: maf_set_error("bottleneck: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
}
