// 3x3 stride-2 pad-1 conv with the WEIGHTS IN REGISTERS (tile_k = 7 of MAF_OP_CONV3X3S2): ConvWrapper / MPRep.conv2 / RepVGGBlock in deploy form
// (yolov6/layers/common.py:76-83, 776-792, 216-217) for the layers whose 9 * Cin * Cout weights do not fit the LDS but do fit the register
// file of one workgroup — 128 -> 128 (the side convs of the MAFPN neck, backbone.23 / .24 / .27 / .28 of MAF-YOLO-n: 295 KB), 96 -> 96, 96 -> 64, 64 -> 64 and
// (round 6) the side convs of s whose input fits the 256-byte patch pixel: 64 -> 96, 128 -> 96.
//
// Why: the generic template (conv_mfma.inc.h VAR_3X3S2) reads every input pixel 2.25 times (the taps of neighbouring outputs overlap) for
// every channel tile and fetches its weight fragments once per wave and k-step: knock-out builds (tools/probe_ko.sh) put 40 of 79 us of the
// twin launch 23 + 24 on the activation loads and 27 us on the weight loads.  Here
//   * a wave OWNS one tile of 16 output channels and keeps ALL of its weight fragments — 9 taps x Cin / 32 k-steps = 144 VGPRs for Cin = 128 —
//     in registers for the whole kernel (one workgroup of 8 waves per CU for 128 -> 128: two waves per SIMD, <= 256 registers each);
//   * the (2 TR + 1) x (2 TC + 1) input patch of a TR x TC = 4 x 8 output tile travels global -> LDS by DMA (global_load_lds, no registers),
//     double-buffered: the patch of the workgroup's next tile is in flight while this one is multiplied, and every input pixel is read once
//     plus a 1.2x halo; out-of-image pixels and channel chunks past Cin read a 16-byte zero page, so the DMA stays unconditional;
//   * the LDS image must be lane-linear for the DMA (wave-uniform base + lane * 16), so bank conflicts are avoided by permuting the SOURCE:
//     pixel (py, px) stores its 16-byte channel chunk c in slot c ^ swz(py, px), swz = ((px >> 1) & 7) | (((py >> 1) & 1) << 3): the 16
//     pixels of an MFMA fragment read (2 output rows x 8 output columns, input stride 2) then hit 16 different slots of the 256-byte bank row;
//   * the product is taken transposed (A = weight fragment from registers: 16 output channels x 32 k; B = patch fragment: one ds_read_b128
//     per lane; K = (tap, 8-channel group) pairs, tap-major, four pairs per k-step — the tap is uniform per k-step because Cin % 32 == 0), so
//     a lane ends up with 4 consecutive output channels of one pixel: bias + activation, 8-byte store into a double-buffered LDS tile, whole
//     NHWC pixels out as 16-byte pieces after the barrier that also publishes the next patch;
//   * a twin launch (two convs of one shape: op->aux) gives each conv half of the workgroups.
#include "maf_common.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko KO_SRCS=conv3s2_wreg.hip): 128 = plain round-robin tile order
#endif
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16w[4];

struct C3wArgs {
    const half_t* in[2];
    const char* rec[2];
    half_t* out[2];
    int nconv, B, Hin, Win, H, W, in_stride, in_coff, out_stride, out_coff, act, tilesX, tilesY, ntiles;   // ntiles per conv
    int out1_coff;                                                              // C1 > 0: channel offset of the pooled branch (out1_coff + C1 == out_coff)
};

constexpr int W3_TR = 4, W3_TC = 8, W3_SR = 2 * W3_TR + 1, W3_SC = 2 * W3_TC + 1, W3_NPIX = W3_SR * W3_SC;
constexpr int W3_SLOTS = W3_NPIX * 16;                              // 16-byte slots of a patch image (256 bytes per pixel, whatever Cin)

// LDS reads as inline assembly with hand-counted waits: written as plain loads the compiler issues every fragment read right in front of the MFMA that
// needs it and waits with lgkmcnt(0) — a full LDS round trip (~130 cycles) per 17-cycle MFMA, 9.4k of the 9.6k cycles a tile took.
template <int OFF> __device__ __forceinline__ void w3_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void w3_wait_lgkm(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }   // "+v": the MFMA that reads `a` cannot move above the wait

template <int N, int I = 0, typename F>
__device__ __forceinline__ void w3_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w3_static_for<N, I + 1>(f);
    }
}

template <int NW>
constexpr int w3_patch_bytes() { return (W3_SLOTS + NW * 64 - 1) / (NW * 64) * (NW * 64) * 16; }   // whole DMA rounds: the last one writes past the last pixel

template <int ACT, int MT, int NTW>
__device__ __forceinline__ void w3_epilogue(const f32x4_t (&acc)[MT][NTW], const f32x4_t (&bv)[NTW], half_t* so, int cout, int ch0, int pix0, int n, int g) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            half4_t v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (half_t)maf_act<ACT>(acc[m][t][q] + bv[t][q]);
            *reinterpret_cast<half4_t*>(so + (pix0 + m * 16 + n) * cout + ch0 + t * 16 + 4 * g) = v;
        }
}

// NWN x NWM waves: wave (wn, wm) owns the COUT / 16 / NWN channel tiles starting at wn * NTW and the 2 / NWM m-tiles (16 pixels = 2 rows x 8 columns)
// starting at wm * MT.  Two waves per SIMD (512-thread workgroups, <= 256 registers per lane): one wave's LDS reads, address arithmetic and epilogue
// run under the other's MFMAs — with one wave per SIMD (a first version: 4 waves holding two channel tiles each) a tile took 14k cycles.
// C1 = COUT: MPRep in one launch (common.py:776-792, cat(conv1(MaxPool2d(2, 2)(x)), conv2(x))): the 2 x 2 window of an output pixel is taps (1, 1), (1, 2),
// (2, 1), (2, 2) of its 3 x 3 window, so the pooled branch's operand is the element-wise maximum of four fragments the conv reads anyway — no extra LDS
// read; its 1x1 + SiLU runs on the same wave -> channel-tile assignment and its C1 channels are stored in front of the conv's (one run per pixel).
template <int CIN, int COUT, int NWN, int NWM, int NBUF, int C1 = 0>
__global__ __launch_bounds__(NWN * NWM * 64, (NWN * NWM + 3) / 4) void conv3s2_wreg_kernel(const C3wArgs a) {
    constexpr int NW = NWN * NWM, NTW = COUT / 16 / NWN, MT = 2 / NWM;
    static_assert(CIN % 32 == 0 && CIN <= 128 && COUT == NWN * NTW * 16 && MT * NWM == 2, "shape");
    static_assert(C1 == 0 || C1 == COUT, "the pooled branch shares the conv's channel tiles");
    constexpr int KPT = CIN / 32, KS = 9 * KPT, GR = CIN / 8, NT = NW * 64, CO2 = COUT + C1;
    constexpr int PB = w3_patch_bytes<NW>(), OB = W3_TR * W3_TC * CO2 * 2;
    constexpr int ROUNDS = (W3_SLOTS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char w3_raw[];       // [NBUF][PB] patches | [2][OB] output tiles
    unsigned char* const s_out = w3_raw + NBUF * PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
    const int wn = wave % NWN, wm = wave / NWN;
    const int wgc = gridDim.x / a.nconv;                                         // workgroups per conv
    const int conv = blockIdx.x / wgc, wg = blockIdx.x - conv * wgc;
    const half_t* const in = a.in[conv];
    half_t* const out = a.out[conv];
    const char* const rec = a.rec[conv];

    // ---- this wave's weight fragments and bias (record: [channel tile][k-step][64 lanes][8])
    half8_t w[KS][NTW];
    {
        const half8_t* wsrc = reinterpret_cast<const half8_t*>(rec) + ((size_t)(wn * NTW) * KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < NTW; ++t) w[s][t] = wsrc[(t * KS + s) * 64];
    }
    const float* bias = reinterpret_cast<const float*>(rec + (size_t)(COUT / 16) * KS * 1024);
    f32x4_t bv[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bv[t] = *reinterpret_cast<const f32x4_t*>(bias + (wn * NTW + t) * 16 + 4 * g);
    // pooled branch: record behind the conv's — fragments [channel tile][CIN / 32 k-steps][64 lanes][8] | bias fp32 [C1]
    half8_t w1[KPT][NTW];
    f32x4_t b1v[NTW];
    if constexpr (C1 > 0) {
        const char* rec1 = rec + (size_t)(COUT / 16) * KS * 1024 + COUT * 4;
        const half8_t* w1src = reinterpret_cast<const half8_t*>(rec1) + ((size_t)(wn * NTW) * KPT) * 64 + lane;
#pragma unroll
        for (int j = 0; j < KPT; ++j)
#pragma unroll
            for (int t = 0; t < NTW; ++t) w1[j][t] = w1src[(t * KPT + j) * 64];
        const float* bias1 = reinterpret_cast<const float*>(rec1 + (size_t)(C1 / 16) * KPT * 1024);
#pragma unroll
        for (int t = 0; t < NTW; ++t) b1v[t] = *reinterpret_cast<const f32x4_t*>(bias1 + (wn * NTW + t) * 16 + 4 * g);
    }

    // ---- patch DMA: slot L of the image <- chunk (L & 15) ^ swz of pixel L >> 4.  What does not depend on the tile is computed once per lane
    // and DMA round: the pixel's place in the patch and its channel chunk (packed: py << 16 | px << 8 | chunk; chunk 31 = a slot nothing maps to)
    int dslot[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int L = wave * 64 + r * NT + lane;
        const int p = L >> 4, slot = L & 15;
        const int py = p / W3_SC, px = p - py * W3_SC;
        const int chunk = slot ^ (((px >> 1) & 7) | (((py >> 1) & 1) << 3));
        dslot[r] = (py << 16) | (px << 8) | ((p < W3_NPIX && chunk < GR) ? chunk : 31);
    }
    auto dma = [&](int tile, int buf) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const half_t* img = in + (size_t)b * a.Hin * a.Win * a.in_stride + a.in_coff;
        const int iy0 = 2 * ty * W3_TR - 1, ix0 = 2 * tx * W3_TC - 1;
        unsigned char* dst = w3_raw + buf * PB + wave * 1024;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            // (every wave issues every round — the slots past the last pixel read the zero page into the padding of the image — so that a
            // counted s_waitcnt vmcnt(ROUNDS) means "everything but the newest patch has landed" for every wave)
            const int d = dslot[r], chunk = d & 255;
            const int iy = iy0 + (d >> 16), ix = ix0 + ((d >> 8) & 255);
            const bool ok = chunk != 31 && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const half_t* src = ok ? img + ((size_t)iy * a.Win + ix) * a.in_stride + 8 * chunk : reinterpret_cast<const half_t*>(g_zero16w);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(dst + r * (NT * 16)), 16, 0, 0);
        }
    };

    // ---- fragment addressing: lane (g, n) = output pixel (row n >> 3, column n & 7) of an m-tile (2 rows x 8 columns), k-pair group 4 (s % KPT) + g
    const int r_ = n >> 3, c_ = n & 7;
    const int pixbase = ((4 * (wm * MT) + 2 * r_) * W3_SC + 2 * c_) * 256;
    int swz[2][2];                                                                // [ky >> 1][kx >> 1]
#pragma unroll
    for (int yk = 0; yk < 2; ++yk)
#pragma unroll
        for (int xk = 0; xk < 2; ++xk) swz[yk][xk] = ((c_ + xk) & 7) | (((r_ + yk) & 1) << 3);

    const int act = a.act;
    // XCD-contiguous tile order (as csrc/stem2.hip): workgroups go to the 8 XCDs round-robin and neighbouring tiles share halo rows / columns — an XCD
    // walks one contiguous eighth of a conv's tiles, so the shared lines are hits in ITS L2 (wgc % 8 == 0: the conv's workgroup index keeps blockIdx's XCD)
    int tile = wg, t_end = a.ntiles, t_step = wgc;
    if ((wgc & 7) == 0 && !(MAF_KO & 128)) {
        const int xcd = wg & 7, q = a.ntiles >> 3, r = a.ntiles & 7, base = xcd * q + min(xcd, r);
        tile = base + (wg >> 3); t_end = base + q + (xcd < r ? 1 : 0); t_step = wgc >> 3;
    }
    const int t_last = t_end > 0 ? t_end - 1 : 0;
    // NBUF - 1 patches in flight ahead of the one being multiplied (NBUF = 3: a patch has two tile times to arrive)
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k)
        if (tile + k * t_step < t_end) dma(tile + k * t_step, k);
        else dma(t_last, k);                                                      // keep the DMA count per wave uniform (the data is never read)
    if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROUNDS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; tile < t_end; ++it, tile += t_step) {
        const int cur = it % NBUF, ocur = it & 1;
        {
            const int nt_ = tile + (NBUF - 1) * t_step;
            dma(nt_ < t_end ? nt_ : t_last, (it + NBUF - 1) % NBUF);              // every wave left that buffer before the last barrier
        }
        // lane addresses of the fragment reads: [ky >> 1][kx >> 1][k-step within the tap] (the rest of an address is a compile-time offset)
        uint32_t fa[2][2][KPT];
        {
            const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(w3_raw + cur * PB + pixbase);
#pragma unroll
            for (int yk = 0; yk < 2; ++yk)
#pragma unroll
                for (int xk = 0; xk < 2; ++xk)
#pragma unroll
                    for (int j = 0; j < KPT; ++j) fa[yk][xk][j] = base + (uint32_t)(((4 * j + g) ^ swz[yk][xk]) * 16);
        }
        f32x4_t acc[MT][NTW];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // KS * MT steps (k-step s, m-tile m), ONE straight line, software-pipelined by hand: the read of step t + RD is issued before the MFMAs of
        // step t (LDS returns in order: when step t is consumed, the min(RD, steps left) reads issued after its own may still be in flight)
        // (C1 > 0: m-tile-major — one m-tile's pooled operand, 4 KPT registers, is complete and consumed before the next one starts — and a shorter read-ahead:
        // with the k-step-major order and RD = 6 the kernel spilled 30 registers)
        constexpr int NSTEP = KS * MT, RD = C1 > 0 ? 4 : 6;
        u32x4_t fr[RD + 1];
        half8_t pool[KPT];                                                        // C1 > 0: running maximum over taps 4, 5, 7, 8 of every channel chunk
        half_t* so = reinterpret_cast<half_t*>(s_out + ocur * OB);                // free since the barrier of the previous tile (its readers copied out two tiles ago)
        auto ld_step = [&](auto idx) {
            constexpr int t = decltype(idx)::value;
            if constexpr (t < NSTEP) {
                constexpr int s_ = C1 > 0 ? t % KS : t / MT, m = C1 > 0 ? t / KS : t % MT, tap = s_ / KPT, ky = tap / 3, kx = tap - 3 * ky;
                constexpr int off = (ky * W3_SC + kx) * 256 + m * (4 * W3_SC * 256);
                static_assert(off < 65536, "ds offset field");
                w3_ds_read_b128<off>(fr[t % (RD + 1)], fa[ky >> 1][kx >> 1][s_ % KPT]);
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // the counter now counts only the reads below
        w3_static_for<RD>([&](auto idx) { ld_step(idx); });
        w3_static_for<NSTEP>([&](auto idx) {
            constexpr int t = decltype(idx)::value, s_ = C1 > 0 ? t % KS : t / MT, m = C1 > 0 ? t / KS : t % MT, sl = t % (RD + 1);
            ld_step(std::integral_constant<int, t + RD>{});
            constexpr int ahead = (NSTEP - 1 - t) < RD ? (NSTEP - 1 - t) : RD;
            w3_wait_lgkm<ahead>(fr[sl]);
            const half8_t f = __builtin_bit_cast(half8_t, fr[sl]);
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) acc[m][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[s_][tt], f, acc[m][tt], 0, 0, 0);
            if constexpr (C1 > 0) {
                constexpr int tap = s_ / KPT, j = s_ % KPT;
                if constexpr (tap == 4) pool[j] = f;
                else if constexpr (tap == 5 || tap == 7 || tap == 8) pool[j] = __builtin_elementwise_max(pool[j], f);   // 4 v_pk_max_f16 (an element-wise ternary is ~30 instructions)
                if constexpr (s_ == KS - 1) {                                     // this m-tile's pooled branch: SiLU(W1 . max + b1) -> channels 0 .. C1 of the staged pixels
#pragma unroll
                    for (int tt = 0; tt < NTW; ++tt) {
                        f32x4_t a1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int jj = 0; jj < KPT; ++jj) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[jj][tt], pool[jj], a1, 0, 0, 0);
                        half4_t v;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = (half_t)maf_act<MAF_ACT_SILU>(a1[q] + b1v[tt][q]);
                        *reinterpret_cast<half4_t*>(so + ((wm * MT + m) * 16 + n) * CO2 + (wn * NTW + tt) * 16 + 4 * g) = v;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                                    // keep the issue order as written
        });
        // ---- bias + activation (picked once per tile, not per value) -> the output tile in LDS: pixel (wm * MT + m) * 16 + n, channels (wn * NTW + t) * 16 + 4 g ..
        if (act == MAF_ACT_SILU) w3_epilogue<MAF_ACT_SILU, MT, NTW>(acc, bv, so, CO2, C1 + wn * NTW * 16, wm * MT * 16, n, g);
        else if (act == MAF_ACT_RELU) w3_epilogue<MAF_ACT_RELU, MT, NTW>(acc, bv, so, CO2, C1 + wn * NTW * 16, wm * MT * 16, n, g);
        else if (act == MAF_ACT_NONE) w3_epilogue<MAF_ACT_NONE, MT, NTW>(acc, bv, so, CO2, C1 + wn * NTW * 16, wm * MT * 16, n, g);
        else w3_epilogue<MAF_ACT_SIGMOID, MT, NTW>(acc, bv, so, CO2, C1 + wn * NTW * 16, wm * MT * 16, n, g);
        // the next patch has landed (this wave's pieces; with three buffers the one after it may still fly) and this wave's part of the output tile is
        // written; the barrier publishes both (a raw s_barrier: __syncthreads() would drain the whole DMA queue with vmcnt(0))
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ROUNDS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const int Y0 = ty * W3_TR, X0 = tx * W3_TC;
        constexpr int CPP = CO2 / 8;
        const int coff = C1 > 0 ? a.out1_coff : a.out_coff;                       // (the pooled channels sit right in front of the conv's)
        for (int q = tid; q < W3_TR * W3_TC * CPP; q += NT) {
            const int px = q / CPP, part = q - px * CPP;
            const int oy = Y0 + (px >> 3), ox = X0 + (px & 7);
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<uint4*>(out + ((size_t)(b * a.H + oy) * a.W + ox) * a.out_stride + coff + 8 * part) =
                    *reinterpret_cast<const uint4*>(so + px * CO2 + 8 * part);
        }
    }
}

template <int CIN, int COUT, int NWN, int NWM, int NBUF, int C1 = 0>
int launch_w3(const C3wArgs& a, int wg_per_conv, hipStream_t s) {
    constexpr int NW = NWN * NWM;
    constexpr int lds = NBUF * w3_patch_bytes<NW>() + 2 * W3_TR * W3_TC * (COUT + C1) * 2;
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s2_wreg_kernel<CIN, COUT, NWN, NWM, NBUF, C1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(conv3s2_wreg)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((conv3s2_wreg_kernel<CIN, COUT, NWN, NWM, NBUF, C1>), dim3(wg_per_conv * a.nconv), dim3(NW * 64), lds, s, a);
    return maf_check_hip(hipGetLastError(), "conv3s2_wreg launch");
}

}  // namespace

// (waves along the channels, waves along the pixels) of the (Cin, Cout) instantiations; 0 = none
static int w3_shape(int cin, int cout, int* nwn, int* nwm) {
    if (cin == 128 && cout == 128) { *nwn = 8; *nwm = 1; return 1; }
    if (cin == 96 && cout == 96) { *nwn = 6; *nwm = 1; return 1; }
    if (cin == 96 && cout == 64) { *nwn = 4; *nwm = 2; return 1; }
    if (cin == 64 && cout == 64) { *nwn = 4; *nwm = 2; return 1; }
    // round 6 — the ConvWrapper side convs of s whose input fits the 256-byte patch pixel (Cin <= 128): backbone.18 / .14 (64 -> 96 on 160 x 160, 128 -> 96 on 80 x 80),
    // one 16-channel tile per wave.  (m's backbone.18, 96 -> 192, was tried as twelve waves of one tile each: 168 registers with spills, two patch buffers only — the
    // generic template stayed faster, 178 us, and the instantiation was dropped.)
    if (cin == 64 && cout == 96) { *nwn = 6; *nwm = 1; return 1; }
    if (cin == 128 && cout == 96) { *nwn = 6; *nwm = 1; return 1; }
    return 0;
}

extern "C" int64_t maf_conv3s2_wreg_record_bytes(int32_t Cin, int32_t Cout) {
    int nwn, nwm;
    if (!w3_shape(Cin, Cout, &nwn, &nwm)) return 0;
    return (int64_t)(Cout / 16) * (9 * Cin / 32) * 1024 + Cout * 4;
}

// ... with the pooled 1x1 branch of MPRep behind it (C1 = Cout = Cin = 96): fragments [C1 / 16][Cin / 32][64][8] f16 | bias fp32 [C1]; 0 = no such kernel
extern "C" int64_t maf_mprep_wreg_record_bytes(int32_t Cin, int32_t Cout, int32_t C1) {
    if (!(Cin == 96 && Cout == 96 && C1 == 96)) return 0;
    return maf_conv3s2_wreg_record_bytes(Cin, Cout) + (int64_t)(C1 / 16) * (Cin / 32) * 1024 + C1 * 4;
}

int maf_launch_conv3s2_wreg(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 && !op->out_f32, "conv3x3s2 (tile_k = 7): fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && sr.C == op->Cin, "conv3x3s2 (tile_k = 7): one direct source");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "conv3x3s2 (tile_k = 7): 16-byte aligned channel slices");
    MAF_REQUIRE(op->w && op->out, "conv3x3s2 (tile_k = 7): null pointer");
    MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->Hin - 1) / 2 + 1 == op->H && (op->Win - 1) / 2 + 1 == op->W, "conv3x3s2: H,W must equal floor((Hin-1)/2)+1");
    MAF_REQUIRE(op->act >= 0 && op->act <= 3, "conv3x3s2: bad act");
    int nwn, nwm;
    if (!w3_shape(op->Cin, op->Cout, &nwn, &nwm)) {
        maf_set_error("conv3x3s2 (tile_k = 7): (Cin, Cout) must be (128, 128), (96, 96), (96, 64), (64, 64), (64, 96) or (128, 96)");
        return MAF_E_UNSUPPORTED;
    }
    C3wArgs a;
    a.in[0] = static_cast<const half_t*>(sr.ptr); a.rec[0] = static_cast<const char*>(op->w); a.out[0] = static_cast<half_t*>(op->out);
    a.nconv = 1;
    if (op->aux[0]) {                                              // twin: {src, record, (unused), out} of a second conv of the same shape
        MAF_REQUIRE(op->aux[1] && op->aux[3], "conv3x3s2 (tile_k = 7) twin: aux = {src, record, -, out}");
        a.in[1] = static_cast<const half_t*>(op->aux[0]); a.rec[1] = static_cast<const char*>(op->aux[1]); a.out[1] = static_cast<half_t*>(const_cast<void*>(op->aux[3]));
        a.nconv = 2;
    } else {
        a.in[1] = a.in[0]; a.rec[1] = a.rec[0]; a.out[1] = a.out[0];
    }
    a.B = op->B; a.Hin = op->Hin; a.Win = op->Win; a.H = op->H; a.W = op->W;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff; a.act = op->act;
    a.tilesX = maf_cdiv(a.W, W3_TC); a.tilesY = maf_cdiv(a.H, W3_TR); a.ntiles = a.B * a.tilesX * a.tilesY;
    // one workgroup per CU (the weights fill the register file): tile_c = workgroups per conv / 32 (default: 256 CUs shared by the convs of the launch)
    int per = op->tile_c > 0 ? op->tile_c * 32 : 256 / a.nconv;
    if (per > a.ntiles) per = a.ntiles;
    const bool two = op->tile_p == 2;                               // tile_p = 2: two patch buffers (A/B); else three
    a.out1_coff = op->reg_stride;
    if (op->nc) {                                                   // MPRep: the MaxPool2d(2, 2) + 1x1 + SiLU branch rides along (nc = its channels, reg_stride = where they go)
        MAF_REQUIRE(op->Cin == 96 && op->Cout == 96 && op->nc == 96 && a.nconv == 1, "conv3x3s2 (tile_k = 7): the pooled 1x1 branch exists for 96 -> 96 + 96, single launches");
        MAF_REQUIRE(op->Hin == 2 * op->H && op->Win == 2 * op->W, "conv3x3s2 (tile_k = 7) with the pooled branch: even input size (MaxPool2d(2, 2) windows)");
        MAF_REQUIRE(op->reg_stride >= 0 && op->reg_stride % 8 == 0 && op->reg_stride + op->nc == op->out_coff, "conv3x3s2 (tile_k = 7): the pooled branch's channels (reg_stride = their offset) must end where the conv's begin");
        return two ? launch_w3<96, 96, 6, 1, 2, 96>(a, per, s) : launch_w3<96, 96, 6, 1, 3, 96>(a, per, s);
    }
#define MAF_W3(CI, CO, NWN_, NWM_) return two ? launch_w3<CI, CO, NWN_, NWM_, 2>(a, per, s) : launch_w3<CI, CO, NWN_, NWM_, 3>(a, per, s)
    if (op->Cin == 128 && op->Cout == 128) MAF_W3(128, 128, 8, 1);
    if (op->Cin == 96 && op->Cout == 96) MAF_W3(96, 96, 6, 1);
    if (op->Cin == 96 && op->Cout == 64) MAF_W3(96, 64, 4, 2);
    if (op->Cin == 64 && op->Cout == 96) MAF_W3(64, 96, 6, 1);
    if (op->Cin == 128 && op->Cout == 96) MAF_W3(128, 96, 6, 1);
    MAF_W3(64, 64, 4, 2);
#undef MAF_W3
}
