// MAF_OP_STEM2: backbone.0 + backbone.1 in one launch — the two RepVGGBlocks in deploy form (3x3 stride-2 conv + bias + ReLU each,
// yolov6/layers/common.py:216-217) that take the 3-channel NCHW image to the 1/4-resolution C1-channel NHWC map, with the `/255` of
// yolov6/core/evaler.py:161-163 folded in for uint8 images.  The 1/2-resolution C0-channel tensor between them is the largest
// activation of the network (32 x 320 x 320 x 24 fp16 = 157 MB written and read back); here it lives in LDS.
//
// One workgroup iteration = a 8 x 16 tile of the final map:
//   A  the 35 x 67 x 3 input patch goes to LDS (planar, zero outside the image);
//   B  the 17 x 33 stem outputs the tile needs are computed on the matrix cores: K = 27 taps (padded to 32) gathered from the patch
//      into fragments (8 two-byte LDS reads per lane), weight fragments in registers, product taken transposed so a lane holds 4
//      consecutive channels of one pixel (8-byte LDS stores), bias + ReLU, fp16, stored pixel-major
//      in LDS with a 56-byte pixel stride (stride-2 reads of 16 lanes then cover all 64 banks); positions outside the stem map are
//      the second conv's zero padding;
//   C  the second conv as an implicit GEMM: K = 9 taps x 24 channels = 27 (tap, 8-channel) pairs, four pairs per MFMA k-step,
//      A fragments = two 8-byte LDS reads, weight fragments from LDS (staged once per workgroup), bias + ReLU;
//   D  the 128 x C1 outputs leave through LDS as 16-byte pieces of whole NHWC pixels.
// Workgroups are persistent (weights staged once).  fp16 engine only; (C0, C1) = (24, 48) [n], (32, 64) [s] or (48, 96) [m: three stem tiles, 4-row tiles only].
#include "maf_common.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko KO_SRCS=stem2.hip): 1 = no stem conv (phase B), 2 = no second conv (phase C), 4 = no output stores, 8 = no image loads, 16 = no third conv, 32 = no patch writes to LDS, 64 = no SiLU
#endif

namespace {

struct S2Args {
    const void* img;          // [B,3,Hin,Win]
    const char* rec;          // pack_stem2 record
    half_t* out;              // [B,H1,W1,out_stride]
    half_t* out2;             // optional: the upper half of the channels as a tensor of its own [B,H1,W1,out2_stride] (RepHDW's x.split((c_, c_), 1), common.py:940)
    int out2_stride;
    int B, Hin, Win, H0, W0, H1, W1, out_stride, out_coff, tilesX, tilesY, ntiles;
    float in_scale;
};

// 4 consecutive input columns (4-element aligned) as fp16
template <typename TI> struct Chunk;
template <> struct Chunk<half_t> {
    typedef u32x2_t raw;
    static __device__ __forceinline__ half4_t cvt(raw v, float) { return *reinterpret_cast<half4_t*>(&v); }
};
template <> struct Chunk<float> {
    typedef f32x4_t raw;
    static __device__ __forceinline__ half4_t cvt(raw v, float) { return half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }
};
template <> struct Chunk<uint8_t> {
    typedef uint32_t raw;
    static __device__ __forceinline__ half4_t cvt(raw v, float s) {
        return half4_t{(half_t)((float)(v & 0xffu) * s), (half_t)((float)((v >> 8) & 0xffu) * s), (half_t)((float)((v >> 16) & 0xffu) * s), (half_t)((float)(v >> 24) * s)};
    }
};

template <typename TI, int C0, int C1, int TY, int C3>
__global__ __launch_bounds__(256, C0 > 32 ? 1 : TY == 8 ? 2 : 3) void stem2_kernel(const S2Args a) {
    constexpr int NT0 = C0 <= 32 ? 2 : 3;                        // 16-channel tiles of the stem conv (m: 48 channels, round 6)
    constexpr int MR = TY / 4;                                   // output rows (16-pixel m-tiles) per wave
    constexpr int TX = 16, SR = 2 * TY + 1, SC = 2 * TX + 1, SP = SR * SC, IR = 2 * SR + 1, IC = 2 * SC + 1;
    constexpr int NCH = (IC + 1 + 3) / 4 + 1, ICS = 4 * NCH;     // patch rows as 18 aligned 4-column chunks starting one column left of the patch
    constexpr int NCHUNK = 3 * IR * NCH, PF = (NCHUNK + 255) / 256;
    constexpr int GR = C0 / 8, NP = 9 * GR, KS1 = (NP + 3) / 4, NT1 = C1 / 16;
    constexpr int TS = (C0 / 2 + 2) | 2;                         // pixel stride of T in dwords: 2 * odd  (24 ch: 14, 32 ch: 18)
    static_assert((TS / 2) % 2 == 1 && TS * 2 >= C0, "T stride");
    constexpr int TSH = TS * 2;                                  // ... in halves
    constexpr int W0B = NT0 * 64 * 16, W1B = KS1 * NT1 * 64 * 16, B0F = 16 * NT0;
    constexpr int KS3 = (C1 + 31) / 32, NT3 = C3 / 16, W3B = KS3 * NT3 * 64 * 16, CO = C3 > 0 ? C3 : C1;   // optional third conv: 1x1, C1 -> C3, SiLU
    constexpr int IN_H = 4 * NCHUNK, OUT_H = TY * TX * CO;
    __shared__ __attribute__((aligned(16))) half_t s_in[(IN_H > OUT_H ? IN_H : OUT_H) + 8];
    __shared__ __attribute__((aligned(16))) half_t s_T[SP * TSH + 64];
    __shared__ __attribute__((aligned(16))) char s_w1[W1B + W3B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.rec + W0B);
        uint4* dst = reinterpret_cast<uint4*>(s_w1);
        for (int i = tid; i < W1B / 16; i += 256) dst[i] = src[i];
        if constexpr (C3 > 0) {                                  // third conv's fragments sit behind the two bias vectors
            const uint4* src3 = reinterpret_cast<const uint4*>(a.rec + W0B + W1B + (B0F + C1) * 4);
            for (int i = tid; i < W3B / 16; i += 256) dst[W1B / 16 + i] = src3[i];
        }
    }
    const half8_t w0a = reinterpret_cast<const half8_t*>(a.rec)[lane], w0b = reinterpret_cast<const half8_t*>(a.rec)[64 + lane];
    const half8_t w0c = reinterpret_cast<const half8_t*>(a.rec)[(NT0 - 1) * 64 + lane];                                               // third tile (NT0 = 3; else = w0b, unused)
    const float* bias = reinterpret_cast<const float*>(a.rec + W0B + W1B);
    const f32x4_t b0a = *reinterpret_cast<const f32x4_t*>(bias + 4 * g), b0b = *reinterpret_cast<const f32x4_t*>(bias + 16 + 4 * g);   // lane (g, n): channels 16t + 4g + q
    const f32x4_t b0c = *reinterpret_cast<const f32x4_t*>(bias + 16 * (NT0 - 1) + 4 * g);
    f32x4_t b1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) b1[t] = *reinterpret_cast<const f32x4_t*>(bias + B0F + 16 * t + 4 * g);
    f32x4_t b3[NT3 > 0 ? NT3 : 1];
    if constexpr (C3 > 0) {
        const float* bias3 = reinterpret_cast<const float*>(a.rec + W0B + W1B + (B0F + C1) * 4 + W3B);
#pragma unroll
        for (int t = 0; t < NT3; ++t) b3[t] = *reinterpret_cast<const f32x4_t*>(bias3 + 16 * t + 4 * g);
    }
    int off0[8];                                                 // patch offsets of this lane's 8 taps (k = 8g + j; k >= 27 meets zero weights)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = min(8 * g + j, 26);
        off0[j] = ((k / 9) * IR + (k % 9) / 3) * ICS + k % 3 + 1;   // + 1: the chunks start one column left of the patch
    }
    int off1[KS1];                                               // T offsets (halves) of this lane's (tap, group) pair in every k-step
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int q = min(4 * s + g, NP - 1), tap = q / GR, grp = q - tap * GR;
        off1[s] = ((tap / 3) * SC + tap % 3) * TSH + 8 * grp;
    }
    const half8_t* w1 = reinterpret_cast<const half8_t*>(s_w1);
    typedef typename Chunk<TI>::raw raw_t;
    raw_t pf[PF];                                                // the next tile's input patch, in flight while this tile computes
    bool pf_in[PF];
    auto prefetch = [&](int tile) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const TI* img = static_cast<const TI*>(a.img) + (size_t)b * 3 * a.Hin * a.Win;
        const int iy0 = 4 * ty * TY - 3, ix0 = 4 * tx * TX - 4;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = min(tid + 256 * u, NCHUNK - 1);
            const int c = e / (IR * NCH), rem = e - c * (IR * NCH), r = rem / NCH, ch = rem - r * NCH;
            const int iy = iy0 + r, ix = ix0 + 4 * ch;
            pf_in[u] = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const size_t at = pf_in[u] ? ((size_t)c * a.Hin + iy) * a.Win + ix : 0;      // clamped: the load stays unconditional
            if (MAF_KO & 8) pf[u] = raw_t{}; else pf[u] = *reinterpret_cast<const raw_t*>(img + at);
        }
    };
    // XCD-contiguous tile order: workgroups go to the 8 XCDs round-robin, and a tile shares the 128-byte lines of its left halo column and its
    // three halo rows with its neighbours — an XCD walks ONE contiguous eighth of the tiles, its resident workgroups a window of consecutive
    // tiles (several tile rows), so those lines are hits in that XCD's L2 instead of a second fetch through another one
    int t_first = blockIdx.x, t_end = a.ntiles, t_step = gridDim.x;
    if ((gridDim.x & 7) == 0 && !(MAF_KO & 128)) {
        const int xcd = blockIdx.x & 7, q = a.ntiles >> 3, r = a.ntiles & 7, base = xcd * q + min(xcd, r);
        t_first = base + (int)(blockIdx.x >> 3); t_end = base + q + (xcd < r ? 1 : 0); t_step = gridDim.x >> 3;
    }
    if (t_first < t_end) prefetch(t_first);
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const int Y0 = ty * TY, X0 = tx * TX;
        __syncthreads();                                         // s_in (= the previous tile's output stage) and s_T are free; first pass: s_w1 is in place
        // ---- A: input patch registers -> LDS (planar, zero outside the image)
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = tid + 256 * u;
            if (e < NCHUNK) {
                half4_t v = Chunk<TI>::cvt(pf[u], a.in_scale);
                if (!pf_in[u]) v = half4_t{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                if (!(MAF_KO & 32)) *reinterpret_cast<half4_t*>(s_in + 4 * e) = v;   // chunk e = (c * IR + r) * NCH + ch sits at that index: rows are ICS = 4 NCH wide
            }
        }
        __syncthreads();
        if (tile + t_step < t_end) prefetch(tile + t_step);
        // ---- B: stem outputs of the tile on the matrix cores, computed transposed (A = weights, B = gathered taps) so that a lane ends
        //      up with 4 consecutive channels of ONE pixel: one 8-byte LDS store per 16-channel tile
        //      Three stages in flight per wave (hipcc orders gather -> wait -> MFMA -> wait -> epilogue per tile: two exposed round trips, nine
        //      times per wave and tile; knock-outs: phases B + C = 29 of 91 us, bound by exactly these waits): the taps of m-tile i + 1 are
        //      gathered and the MFMAs of m-tile i issued BEFORE the bias / ReLU / store work of m-tile i - 1.
        constexpr bool PIPE_B = C0 == 24;      // (32, 64): the extra live fragments push the kernel past 256 registers (one workgroup per CU): the plain loop below
        if constexpr (!PIPE_B) {
            for (int mt = wave; mt < ((MAF_KO & 1) ? 0 : (SP + 15) / 16); mt += 4) {
                const int pp = mt * 16 + n, p = min(pp, SP - 1), r = p / SC, c = p - r * SC;
                const half_t* base = s_in + (2 * r) * ICS + 2 * c;
                half8_t bf;
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = base[off0[j]];
                const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                const f32x4_t ca = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0a, bf, z, 0, 0, 0);
                const f32x4_t cb = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b, bf, z, 0, 0, 0);
                f32x4_t cc = z;
                if constexpr (NT0 == 3) cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0c, bf, z, 0, 0, 0);
                if (pp < SP) {
                    const bool in = (unsigned)(2 * Y0 - 1 + r) < (unsigned)a.H0 && (unsigned)(2 * X0 - 1 + c) < (unsigned)a.W0;   // else: zero padding of conv 2
                    half4_t va, vb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        va[q] = (half_t)(in ? fmaxf(ca[q] + b0a[q], 0.f) : 0.f);
                        vb[q] = (half_t)(in ? fmaxf(cb[q] + b0b[q], 0.f) : 0.f);
                    }
                    *reinterpret_cast<half4_t*>(s_T + pp * TSH + 4 * g) = va;
                    if (16 + 4 * g < C0) *reinterpret_cast<half4_t*>(s_T + pp * TSH + 16 + 4 * g) = vb;
                    if constexpr (NT0 == 3) {
                        half4_t vc;
#pragma unroll
                        for (int q = 0; q < 4; ++q) vc[q] = (half_t)(in ? fmaxf(cc[q] + b0c[q], 0.f) : 0.f);
                        if (32 + 4 * g < C0) *reinterpret_cast<half4_t*>(s_T + pp * TSH + 32 + 4 * g) = vc;
                    }
                }
            }
        } else if (!(MAF_KO & 1)) {
            constexpr int NMT = (SP + 15) / 16, NI = (NMT + 3) / 4;
            half8_t bfr[2];
            f32x4_t car[2], cbr[2];
            auto gather = [&](int mt, half8_t& bf) {
                const int p = min(mt * 16 + n, SP - 1), r = p / SC, c = p - r * SC;
                const half_t* base = s_in + (2 * r) * ICS + 2 * c;
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = base[off0[j]];
            };
            auto finish = [&](int mt, const f32x4_t& ca, const f32x4_t& cb) {
                const int pp = mt * 16 + n, p = min(pp, SP - 1), r = p / SC, c = p - r * SC;
                if (pp < SP) {
                    const bool in = (unsigned)(2 * Y0 - 1 + r) < (unsigned)a.H0 && (unsigned)(2 * X0 - 1 + c) < (unsigned)a.W0;   // else: zero padding of conv 2
                    half4_t va, vb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        va[q] = (half_t)(in ? fmaxf(ca[q] + b0a[q], 0.f) : 0.f);
                        vb[q] = (half_t)(in ? fmaxf(cb[q] + b0b[q], 0.f) : 0.f);
                    }
                    *reinterpret_cast<half4_t*>(s_T + pp * TSH + 4 * g) = va;
                    if (16 + 4 * g < C0) *reinterpret_cast<half4_t*>(s_T + pp * TSH + 16 + 4 * g) = vb;
                }
            };
            gather(wave, bfr[0]);
#pragma unroll
            for (int i = 0; i <= NI; ++i) {                                  // stage i: gather i + 1 | MFMAs i | finish i - 1
                const int mt = wave + 4 * i;
                if (i + 1 < NI && mt + 4 < NMT) gather(mt + 4, bfr[(i + 1) & 1]);
                if (i < NI && mt < NMT) {
                    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    car[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0a, bfr[i & 1], z, 0, 0, 0);
                    cbr[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b, bfr[i & 1], z, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i > 0 && mt - 4 < NMT) finish(mt - 4, car[(i - 1) & 1], cbr[(i - 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- C: second conv, implicit GEMM over (tap, 8-channel group) pairs, transposed as well (A = weight fragments, B = T)
        f32x4_t acc[MR][NT1];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ((MAF_KO & 2) ? 0 : KS1); ++s) {
            half8_t tf[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int yy = wave * MR + m;
                const half_t* tp = s_T + ((2 * yy) * SC + 2 * n) * TSH + off1[s];
                const half4_t lo = *reinterpret_cast<const half4_t*>(tp), hi = *reinterpret_cast<const half4_t*>(tp + 4);
                tf[m] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const half8_t wf = w1[(s * NT1 + t) * 64 + lane];
#pragma unroll
                for (int m = 0; m < MR; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, tf[m], acc[m][t], 0, 0, 0);
            }
        }
        // ---- D: bias + ReLU (-> third conv: 1x1 + SiLU, see below) -> LDS (aliases the input patch: its last reader was phase B) -> whole NHWC pixels
        half_t* s_out = s_in;
        if constexpr (C3 == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    half4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (half_t)fmaxf(acc[m][t][q] + b1[t][q], 0.f);
                    *reinterpret_cast<half4_t*>(s_out + ((wave * MR + m) * TX + n) * C1 + 16 * t + 4 * g) = v;
                }
        } else {
            // The RepHDW block that follows starts with a 1x1 conv on exactly this tensor (backbone.2.conv1): a lane holds channels
            // 16t + 4g + q of pixel n for every tile t, which IS a fragment (pixel column, 8 k-slots) of the next matrix product once that
            // conv's weights are packed with their K axis in this order — so the quarter-resolution C1-channel tensor never exists in memory.
            const half8_t* w3 = reinterpret_cast<const half8_t*>(s_w1 + W1B);
            f32x4_t acc3[MR][NT3];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                half8_t a2[KS3];
#pragma unroll
                for (int j = 0; j < KS3; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a2[j][q] = (half_t)fmaxf(acc[m][2 * j][q] + b1[2 * j][q], 0.f);
                        a2[j][4 + q] = 2 * j + 1 < NT1 ? (half_t)fmaxf(acc[m][(2 * j + 1) % NT1][q] + b1[(2 * j + 1) % NT1][q], 0.f) : (half_t)0.f;
                    }
#pragma unroll
                for (int t = 0; t < NT3; ++t) {
                    acc3[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < ((MAF_KO & 16) ? 0 : KS3); ++j) acc3[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[(t * KS3 + j) * 64 + lane], a2[j], acc3[m][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT3; ++t) {
                    half4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (MAF_KO & 64) ? (half_t)(acc3[m][t][q] + b3[t][q]) : (half_t)maf_act<MAF_ACT_SILU>(acc3[m][t][q] + b3[t][q]);
                    *reinterpret_cast<half4_t*>(s_out + ((wave * MR + m) * TX + n) * C3 + 16 * t + 4 * g) = v;
                }
        }
        // a wave copies out the MR rows it staged itself (its LDS operations execute in order): no workgroup barrier here; the patch this stage
        // aliases was last read in phase B, a barrier ago for every wave, and the barrier at the top of the tile loop keeps the next tile's patch
        // writes behind these reads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int CPP = CO / 8;                               // 16-byte pieces per pixel
        for (int ql = lane; ql < MR * TX * CPP; ql += 64) {
            const int px = wave * MR * TX + ql / CPP, part = ql % CPP;
            const int oy = Y0 + px / TX, ox = X0 + px % TX;
            if (oy < a.H1 && ox < a.W1 && !(MAF_KO & 4)) {
                const size_t pix = (size_t)(b * a.H1 + oy) * a.W1 + ox;
                half_t* dst = a.out + pix * a.out_stride + a.out_coff + 8 * part;
                if (a.out2 && part >= CPP / 2) dst = a.out2 + pix * a.out2_stride + 8 * (part - CPP / 2);   // two dense halves: whole lines for the slice readers
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(s_out + px * CO + 8 * part);
            }
        }
    }
}

}  // namespace

extern "C" int64_t maf_stem2_record_bytes(int32_t C0, int32_t C1, int32_t C3) {
    const int ks1 = (9 * (C0 / 8) + 3) / 4, nt0 = C0 <= 32 ? 2 : 3;
    const int64_t base = nt0 * 64 * 16 + (int64_t)ks1 * (C1 / 16) * 64 * 16 + nt0 * 16 * 4 + C1 * 4;
    return C3 > 0 ? base + (int64_t)((C1 + 31) / 32) * (C3 / 16) * 64 * 16 + C3 * 4 : base;
}

int maf_launch_stem2(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "stem2: fp16 engine only");
    MAF_REQUIRE(op->Cin == 3 && ((op->ksize == 24 && op->Cout == 48) || (op->ksize == 32 && op->Cout == 64) || (op->ksize == 48 && op->Cout == 96)),
                "stem2: (C0, C1) must be (24, 48), (32, 64) or (48, 96); ksize carries C0");
    MAF_REQUIRE(op->act == MAF_ACT_RELU, "stem2: both RepVGG blocks end in ReLU (common.py:198)");
    MAF_REQUIRE(op->nc == 0 || op->nc == op->Cout, "stem2: the optional third conv (nc = its output channels; 1x1 + SiLU) must keep the channel count (48 -> 48, 64 -> 64)");
    MAF_REQUIRE(op->src[0].ptr && op->w && op->out, "stem2: null pointer");
    MAF_REQUIRE(op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "stem2: out stride/coff multiples of 8");
    S2Args a;
    a.img = op->src[0].ptr; a.rec = static_cast<const char*>(op->w); a.out = static_cast<half_t*>(op->out);
    a.B = op->B; a.Hin = op->Hin; a.Win = op->Win;
    a.H0 = (op->Hin - 1) / 2 + 1; a.W0 = (op->Win - 1) / 2 + 1;
    a.H1 = (a.H0 - 1) / 2 + 1; a.W1 = (a.W0 - 1) / 2 + 1;
    MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && op->Win % 4 == 0 && a.H1 == op->H && a.W1 == op->W, "stem2: H,W must be the twice-halved image size, image width a multiple of 4");
    a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.out2 = static_cast<half_t*>(const_cast<void*>(op->aux[0])); a.out2_stride = op->reg_stride;
    MAF_REQUIRE(!a.out2 || (op->nc == op->Cout && op->Cout % 16 == 0 && op->reg_stride % 8 == 0 && op->reg_stride >= op->Cout / 2),
                "stem2: a second output (aux[0] = the upper half of the channels, reg_stride = its pixel stride, a multiple of 8) needs the third conv");
    const int ty_rows = (op->tile_p == 4 || op->Cout == 96) ? 4 : 8;                  // tile height of the quarter-resolution map (tile_p: 0 / 8 = 8 rows, 4 = 4 rows: less LDS and registers, more halo)
    a.tilesX = maf_cdiv(a.W1, 16); a.tilesY = maf_cdiv(a.H1, ty_rows); a.ntiles = a.B * a.tilesX * a.tilesY;
    a.in_scale = op->in_dtype == MAF_U8 ? 1.0f / 255.0f : 1.0f;
    const dim3 grid(std::min(a.ntiles, op->tile_k > 0 ? op->tile_k : op->Cout == 96 ? 256 : (ty_rows == 4 ? 768 : 512))), blk(256);
#define MAF_S2(TI, C0, C1, C3) do { if (ty_rows == 4) hipLaunchKernelGGL((stem2_kernel<TI, C0, C1, 4, C3>), grid, blk, 0, s, a); else hipLaunchKernelGGL((stem2_kernel<TI, C0, C1, 8, C3>), grid, blk, 0, s, a); } while (0)
    // (48, 96) [m, round 6]: 84 + 18 KiB of weights + the 4-row stem tile (31 KiB) + the patch / output stage (12 KiB) = 145 KiB of LDS: 4-row tiles only, one workgroup per CU
#define MAF_S2M(TI) do { if (op->nc) hipLaunchKernelGGL((stem2_kernel<TI, 48, 96, 4, 96>), grid, blk, 0, s, a); else hipLaunchKernelGGL((stem2_kernel<TI, 48, 96, 4, 0>), grid, blk, 0, s, a); } while (0)
#define MAF_S2T(TI) do { if (op->Cout == 96) MAF_S2M(TI); else if (op->Cout == 48) { if (op->nc) MAF_S2(TI, 24, 48, 48); else MAF_S2(TI, 24, 48, 0); } else { if (op->nc) MAF_S2(TI, 32, 64, 64); else MAF_S2(TI, 32, 64, 0); } } while (0)
    if (op->in_dtype == MAF_F16) MAF_S2T(half_t);
    else if (op->in_dtype == MAF_F32) MAF_S2T(float);
    else if (op->in_dtype == MAF_U8) MAF_S2T(uint8_t);
    else { maf_set_error("stem2: bad in_dtype"); return MAF_E_ARG; }
#undef MAF_S2T
#undef MAF_S2M
#undef MAF_S2
    return maf_check_hip(hipGetLastError(), "stem2 launch");
}
