// Pointwise (1x1) and dense 3x3 stride-2 convolutions as MFMA implicit GEMMs, NHWC, fused
// bias + activation epilogue, concat/split/upsample/maxpool folded into the operand addressing.
//
// Replaces (reference, /root/reference): Conv.forward_fuse (yolov6/layers/common.py:49-50),
// RepVGGBlock deploy forward (common.py:216-217), ConvWrapper (common.py:76-83), the cls/reg pred
// convs of Head_DepthUni (common.py:1331-1335), torch.cat (common.py:154,944), Tensor.split
// (common.py:940), nn.Upsample (MAF-YOLO-n.yaml:21,26), MP (common.py:667-673).
//
// Formulation (MI355X-first, not a cuDNN-style tiling):
//   out[m, c] = sum_k X[m, k] * W[k, c]      m = pixel (B*H*W), c = output channel, k = input channel (x tap)
// The ACTIVATIONS are the MFMA A operand: with NHWC storage an A fragment is one 16-byte load per lane
// straight from global memory (8 f16 / 4 f32 consecutive channels of the lane's pixel) — no LDS, no
// im2col.  The WEIGHTS are the B operand, pre-packed on the host in fragment order (1 KiB contiguous
// per wave-load, L1/L2 resident) with the output channels of the CT column tiles interleaved:
// accumulator column p of tile ct is channel p*CT + ct.  Lane (g, p) therefore owns CT *consecutive*
// channels of pixels g*4 .. g*4+3, the 16 lanes of a pixel cover 16*CT consecutive channels, and each
// store instruction of the epilogue writes whole contiguous NHWC rows (4 pixels x 16*CT channels per
// wave-store) straight from the accumulators — full cache lines, no LDS transpose.  (A first version
// with weights as the A operand stored 16-byte pieces 48-64 B apart; the L2 write-request rate, not
// HBM, bounded the wide layers.)
//
// Per wave: 16*PT pixels x 16*CT channels; 4 waves per workgroup split the pixel range.
// blockIdx -> (pixel tile, channel tile) is XCD-aware: the channel tiles of one pixel tile run
// back-to-back on the same XCD so re-reads of the activations hit that XCD's L2.
#pragma once
#include "maf_common.h"
#include <type_traits>

struct ConvArgs {
    const void* src[4];
    int srcStride[4];
    int srcCoff[4];
    int srcMode[4];
    int srcC[4];
    int cum[5];          // cumulative k-step boundaries of the sources (each source padded to whole k-steps)
    int nsrc;
    int B, H, W, Hin, Win;
    int M;               // B*H*W
    int Cin, Cout;
    int ksteps;          // k-steps per tap
    const void* w;
    const float* bias;   // padded to nN*16*CT
    void* out;
    int out_stride, out_coff;
    int nM, nN;
    int act;
    int stream_waves;    // conv1x1_stream_lds only: 4, or 8 (op->tile_p = 2: 512-thread workgroups, csrc/conv_stream_lds_w8.hip)
    int out_pairs;       // conv1x1_stream_lds only: the output is stored as pixel pairs (MAF_SRC_PAIRS) for a depth-wise consumer
    float* stats;        // conv1x1_stream_lds, training (csrc/conv_stream_lds_st.hip): the BatchNorm behind the conv reads its batch statistics from here —
    int stats_R;         //   [stats_R][2][Cout] fp32 {sum, sum of squares} of the ROUNDED outputs, added to by every workgroup (replica = blockIdx % stats_R); null: none
    int dg_mc, dg_nmc;   // VAR_DGRAD3 with even H, W: pixels per parity class (B * H/2 * W/2) and pixel tiles per class (nM = 4 * dg_nmc); 0 = unsplit
    // twin launch (op->aux[0..3]): a SECOND conv of identical shape — same everything except these four pointers — runs as
    // blockIdx.y = 1 of the same grid (the two side convs of a MAFPN level, backbone.23 / .24 and .27 / .28: independent, equal, each too
    // small to fill the chip and each paying a kernel's fixed cost on its own)
    const void* src_t;
    const void* w_t;
    const float* bias_t;
    void* out_t;
    int twin;
};

// VAR_DGRAD3: the DATA GRADIENT of the 3x3 stride-2 pad-1 conv (training, MAF_OP_CONV3X3S2_DGRAD): output pixel = input pixel (iy, ix)
// of the forward conv, source = dY on the half-resolution grid; tap (ky, kx) contributes dY[(iy + 1 - ky) / 2, (ix + 1 - kx) / 2] when both
// are even and in range, else the zero page — 2.25 of the 9 taps on average.  With even H, W (every layer of a 32-aligned image) the pixels
// are therefore walked PARITY CLASS by parity class (a.dg_mc: workgroups of class (y & 1, x & 1) see only pixels of that class) and a
// class runs only ITS taps — 1, 2, 2 or 4 of the 9, i.e. a quarter of the k-steps, loads and MFMAs of the all-taps form.  The gather
// form needs no atomics and no scatter pass.
enum { VAR_DIRECT = 0, VAR_MULTI = 1, VAR_POOL2 = 2, VAR_3X3S2 = 3, VAR_DGRAD3 = 4 };

int maf_conv_mfma_f16(const ConvArgs& a, int var, bool outf32, int pt, int ct, hipStream_t s);
int maf_conv_mfma_f32(const ConvArgs& a, int var, bool outf32, int pt, int ct, hipStream_t s);
int maf_conv_mfma_f16_lb(const ConvArgs& a, int var, int pt, int ct, hipStream_t s);   // weights shared through LDS (tile_k = 2)
int maf_conv_mfma_f16_dma(const ConvArgs& a, int var, int pt, int ct, hipStream_t s);  // ... by DMA, a ring of two-k-step stages (tile_k = 8)
int maf_conv_mfma_dgrad3(const ConvArgs& a, int dtype, int pt, int ct, hipStream_t s);  // VAR_DGRAD3 (conv_mfma_dgrad.hip)
int maf_conv1x1_stream(const ConvArgs& a, int pt, int ct, hipStream_t s);              // persistent waves, cross-tile prefetch (tile_k = 3)
int maf_conv1x1_stream_lds(const ConvArgs& a, int var, int ct, hipStream_t s);        // the same with LDS-resident weights (tile_k = 5)
int maf_conv1x1_stream_lds_st(const ConvArgs& a, int ct, hipStream_t s);              // ... that also accumulates the statistics of the BatchNorm behind it (a.stats)

// Profiling builds only (make ko KO=<bits>, tools/conv_probe.py; never the shipped library): MAF_KO knocks one piece out of conv_mfma_kernel
// to see what it costs.  1: weight fragments are constants (no weight loads)  2: activation fragments are constants (no activation loads)
// 4: no output stores  8: no MFMAs (operands folded into the accumulator with one add, so the loads stay live)
#ifndef MAF_KO
#define MAF_KO 0
#endif

namespace {

// 16 zero bytes in global memory: the stride-2 3x3 conv points its out-of-image taps here, so every operand load is
// unconditional (a load under a divergent branch makes the compiler wait with vmcnt(0), i.e. serialises the whole
// software pipeline: the prefetched fragments of the next k-steps could never stay in flight across the MFMAs)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

template <typename T> struct Frag;
template <> struct Frag<half_t> {
    typedef half8_t type;
    static constexpr int CH = 8;    // channels per 16-byte lane chunk
    static constexpr int KS = 32;   // channels per k-step
    static __device__ __forceinline__ type zero() { return (type)(half_t)0; }
    static __device__ __forceinline__ type vmax(type a, type b) { return __builtin_elementwise_max(a, b); }   // 4 v_pk_max_f16
    static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Frag<float> {
    typedef f32x4_t type;
    static constexpr int CH = 4;
    static constexpr int KS = 16;
    static __device__ __forceinline__ type zero() { return (type)0.f; }
    static __device__ __forceinline__ type vmax(type a, type b) {
        type r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
        return r;
    }
    static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};

template <int OFF> __device__ __forceinline__ void cm_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void cm_wait_lgkm(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }   // "+v": the MFMA that reads `a` stays below the wait
template <int N> __device__ __forceinline__ void cm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N, int I = 0, typename Fn>
__device__ __forceinline__ void cm_static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cm_static_for<N, I + 1>(f);
    }
}

template <typename T>
__device__ __forceinline__ typename Frag<T>::type ldg16(const T* p) {
    return *reinterpret_cast<const typename Frag<T>::type*>(p);
}

template <typename T, int PT, int CT, int VAR, bool OUTF32, bool KS4 = false, bool LB = false, bool DMA = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    static_assert(!KS4 || PT == 1, "split-K variant is defined for one pixel tile per wave");
    static_assert(!(KS4 && LB), "split-K and LDS-shared weights are separate variants");
    static_assert(!DMA || (LB && sizeof(T) == 2 && CT % 2 == 0), "the DMA ring is a form of the LDS-shared-weight variant (fp16, even tile_c)");
    typedef Frag<T> F;
    typedef typename F::type frag_t;
    constexpr int CH = F::CH;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;

    // XCD-aware block -> tile map (block b runs on XCD b%8; keep a pixel tile's channel tiles on one XCD)
    const int L = blockIdx.x;
    const int xcd = L & 7, jj = L >> 3;
    const int n_tile = jj % a.nN;
    const int m_tile = (jj / a.nN) * 8 + xcd;
    if (m_tile >= a.nM) return;
    // KS4: the 4 waves of the workgroup share ONE 16-pixel tile and each takes a quarter of the k-steps (small maps with
    // long reductions are dependent-load chains; this cuts the chain 4x and puts 4x more workgroups on the chip)
    // VAR_DGRAD3, split form: this workgroup's parity class and its tile inside the class
    const bool dgs = VAR == VAR_DGRAD3 && a.dg_mc > 0;
    const int dg_cls = dgs ? m_tile / a.dg_nmc : 0, dg_py = dg_cls >> 1, dg_px = dg_cls & 1;
    const int dg_nky = 1 + dg_py, dg_nkx = 1 + dg_px;                     // odd rows / columns collect two tap rows / columns, even ones the centre
    const int m_base = KS4 ? m_tile * 16 : (dgs ? m_tile - dg_cls * a.dg_nmc : m_tile) * (64 * PT) + wave * (16 * PT);

    // ---- per-lane pixel bookkeeping ----
    bool pvalid[PT];
    uint32_t off0[PT];                      // VAR_DIRECT / VAR_POOL2 / VAR_3X3S2 base element offset
    uint32_t offs[PT][4];                   // VAR_MULTI: per source
    int iy0[PT], ix0[PT];                   // VAR_3X3S2
    const bool second = blockIdx.y != 0;                                  // twin launch: the second conv of the pair
    const T* s0 = static_cast<const T*>(second ? a.src_t : a.src[0]);
    const void* const w_all = second ? a.w_t : a.w;
    const float* const bias_all = second ? a.bias_t : a.bias;
    void* const out_all = second ? a.out_t : a.out;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m_base + pt * 16 + p;
        pvalid[pt] = m < (dgs ? a.dg_mc : a.M);
        const int mm = pvalid[pt] ? m : 0;
        if (VAR == VAR_DIRECT) {
            off0[pt] = (uint32_t)mm * a.srcStride[0] + a.srcCoff[0];
        } else {
            int x, y, b;
            if (dgs) {                                                    // mm counts the pixels of ONE parity class
                const int w2 = a.W >> 1, h2 = a.H >> 1;
                const int t = mm / w2;
                x = 2 * (mm - t * w2) + dg_px;
                y = 2 * (t % h2) + dg_py;
                b = t / h2;
            } else {
                x = mm % a.W;
                const int t = mm / a.W;
                y = t % a.H;
                b = t / a.H;
            }
            if (VAR == VAR_POOL2) {
                off0[pt] = (uint32_t)((b * (2 * a.H) + 2 * y) * (2 * a.W) + 2 * x) * a.srcStride[0] + a.srcCoff[0];
            } else if (VAR == VAR_3X3S2) {
                iy0[pt] = 2 * y - 1;
                ix0[pt] = 2 * x - 1;
                off0[pt] = (uint32_t)(b * a.Hin) * a.Win;       // pixel index of (b, 0, 0)
            } else if (VAR == VAR_DGRAD3) {
                iy0[pt] = y + 1;
                ix0[pt] = x + 1;
                off0[pt] = (uint32_t)(b * a.Hin) * a.Win;       // pixel index of (b, 0, 0) on the dY grid
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < a.nsrc) {
                        if (a.srcMode[s] == MAF_SRC_UP2)
                            offs[pt][s] = (uint32_t)((b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * a.srcStride[s] + a.srcCoff[s];
                        else
                            offs[pt][s] = (uint32_t)mm * a.srcStride[s] + a.srcCoff[s];
                    } else {
                        offs[pt][s] = 0;
                    }
                }
            }
        }
    }

    f32x4_t acc[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = (f32x4_t)0.f;

    const int ntaps = dgs ? dg_nky * dg_nkx : (VAR == VAR_3X3S2 || VAR == VAR_DGRAD3) ? 9 : 1;
    const int total_steps = ntaps * a.ksteps;
    const int w_steps = (VAR == VAR_DGRAD3 ? 9 : ntaps) * a.ksteps;        // k-steps of the packed weights (all nine taps)
    const int s_begin = KS4 ? (total_steps * wave) / 4 : 0;
    const int s_end = KS4 ? (total_steps * (wave + 1)) / 4 : total_steps;
    const frag_t* wbase = reinterpret_cast<const frag_t*>(w_all) + ((size_t)(n_tile * CT) * w_steps) * 64 + lane;

    // Activation fragments of k-step `step` (0 .. last; a step past `last` is reduction padding and reads zeros).
    // EVERY load is unconditional: a load under a branch makes the compiler wait with vmcnt(0) at the next use, which
    // serialises the software pipeline.  So: pixels past the last one read pixel 0 (their rows are never stored), channel
    // chunks past a source's last channel read chunk 0 (their weight rows are zero), out-of-image taps and padding steps
    // read the 16-byte zero page, and the source of a concat is chosen with scalar selects, not branches.
    const int last_step = (KS4 ? (total_steps * (wave + 1)) / 4 : total_steps) - 1;
    const T* zpage = reinterpret_cast<const T*>(g_zero16);
    auto load_b = [&](int step, frag_t (&bf)[PT]) {
        if (MAF_KO & 2) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) bf[pt] = (frag_t)(T)(0.001f * (float)(lane + step));
            return;
        }
        const bool pad = step > last_step;                                 // uniform
        const int sc = pad ? last_step : step;
        if (VAR == VAR_3X3S2) {
            const int tap = sc / a.ksteps, ks = sc - tap * a.ksteps;
            const int ky = tap / 3, kx = tap - ky * 3;
            int c0 = ks * F::KS + g * CH;
            c0 = c0 < a.Cin ? c0 : 0;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int iy = iy0[pt] + ky, ix = ix0[pt] + kx;
                const bool ok = !pad && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;    // else: the zero padding
                const uint32_t o = (off0[pt] + (uint32_t)(iy * a.Win + ix)) * a.srcStride[0] + a.srcCoff[0] + c0;
                bf[pt] = ldg16<T>(ok ? s0 + o : zpage);
            }
        } else if (VAR == VAR_DGRAD3) {
            const int tap = sc / a.ksteps, ks = sc - tap * a.ksteps;
            int ky, kx;
            if (dgs) {                                                    // the class's own taps: both parities match by construction
                const int ty = tap / dg_nkx, tx = tap - ty * dg_nkx;
                ky = dg_py ? 2 * ty : 1;
                kx = dg_px ? 2 * tx : 1;
            } else {
                ky = tap / 3;
                kx = tap - ky * 3;
            }
            int c0 = ks * F::KS + g * CH;
            c0 = c0 < a.Cin ? c0 : 0;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int sy = iy0[pt] - ky, sx = ix0[pt] - kx;                      // twice the dY row / column this tap reads
                const bool ok = !pad && ((sy | sx) & 1) == 0 && (unsigned)(sy >> 1) < (unsigned)a.Hin && (unsigned)(sx >> 1) < (unsigned)a.Win;
                const uint32_t o = (off0[pt] + (uint32_t)((sy >> 1) * a.Win + (sx >> 1))) * a.srcStride[0] + a.srcCoff[0] + c0;
                bf[pt] = ldg16<T>(ok ? s0 + o : zpage);
            }
        } else if (VAR == VAR_MULTI) {
            // every source owns whole k-steps: source index and its first step by scalar selects
            const int i1 = sc >= a.cum[1], i2 = sc >= a.cum[2], i3 = sc >= a.cum[3];
            const int first = i3 ? a.cum[3] : i2 ? a.cum[2] : i1 ? a.cum[1] : 0;
            const int srcC = i3 ? a.srcC[3] : i2 ? a.srcC[2] : i1 ? a.srcC[1] : a.srcC[0];
            const T* sp = static_cast<const T*>(i3 ? a.src[3] : i2 ? a.src[2] : i1 ? a.src[1] : a.src[0]);
            int c0 = (sc - first) * F::KS + g * CH;
            c0 = c0 < srcC ? c0 : 0;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const uint32_t o = i3 ? offs[pt][3] : i2 ? offs[pt][2] : i1 ? offs[pt][1] : offs[pt][0];
                bf[pt] = ldg16<T>(pad ? zpage : sp + o + c0);
            }
        } else {
            int c0 = sc * F::KS + g * CH;
            c0 = c0 < a.Cin ? c0 : 0;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const T* q = pad ? zpage : s0 + off0[pt] + c0;
                if (VAR == VAR_POOL2) {
                    // MAF_SRC_SUB2 (a 1x1 stride-2 conv: RepVGGBlock.rbr_1x1, common.py:203): the same addressing with the window collapsed onto its top-left pixel
                    const bool one = pad || a.srcMode[0] == MAF_SRC_SUB2;
                    const uint32_t cs = one ? 0u : (uint32_t)a.srcStride[0], rs = one ? 0u : (uint32_t)(2 * a.W) * a.srcStride[0];
                    frag_t v0 = ldg16<T>(q), v1 = ldg16<T>(q + cs);
                    frag_t v2 = ldg16<T>(q + rs), v3 = ldg16<T>(q + rs + cs);
                    bf[pt] = F::vmax(F::vmax(v0, v1), F::vmax(v2, v3));
                } else {
                    bf[pt] = ldg16<T>(q);
                }
            }
        }
    };
    auto load_a = [&](int step, frag_t (&af)[CT]) {                       // weights: padding steps re-read the last step (times zero activations)
        if (MAF_KO & 1) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) af[ct] = (frag_t)(T)(0.002f * (float)(lane + ct));
            return;
        }
        int sc = step > last_step ? last_step : step;
        if (dgs) {                                                        // class tap -> its place among the nine packed taps
            const int tap = sc / a.ksteps, ks = sc - tap * a.ksteps;
            const int ty = tap / dg_nkx, tx = tap - ty * dg_nkx;
            sc = ((dg_py ? 2 * ty : 1) * 3 + (dg_px ? 2 * tx : 1)) * a.ksteps + ks;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) af[ct] = wbase[((size_t)ct * w_steps + sc) * 64];
    };
    auto mma_stage = [&](const frag_t (&bf)[PT], const frag_t (&af)[CT]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (MAF_KO & 8) acc[pt][ct][0] += (float)bf[pt][0] + (float)af[ct][0];
                else acc[pt][ct] = F::mma(bf[pt], af[ct], acc[pt][ct]);   // A = activations (rows = pixels), B = weights
            }
    };

    if constexpr (LB && DMA) {
        // tile_k = 8 (round 6): the LDS-shared weights of the branch below, but the fragments travel global -> LDS by DMA (global_load_lds: no registers, no
        // ds_write), in STAGES of two k-steps through a ring of four slots, three stages in flight ahead of the one being multiplied, ONE barrier per stage
        // (2 x PT x CT MFMAs per wave between barriers, 16 .. 32; the branch below: one barrier and one register -> LDS copy per k-step).  The wave's own
        // activation fragments ride the same distance ahead in a register ring.  What it is for: the K-heavy layers of the 40 x 40 / 20 x 20 maps of s and m
        // (1x1 over 960 .. 1536 concatenated channels, 3x3 stride 2 over 192 .. 384), where the generic form streams every wave's weight fragments out of the L2
        // separately (15 TB/s of L2 traffic for 1280 -> 384 on 40 x 40: the bound) and the branch below spends its time at barriers.
        // Knock-outs (make ko, tools/conv_probe.py; 1152 -> 384 on 40 x 40, tile (2, 8)): 76 us as shipped, 60 without the DMA, 55 without the activation loads, 49
        // with neither — 0.96 PFLOP/s is what barriers + fragment reads + MFMAs reach here; the vendor GEMM (tools/gemm_probe.py) does this shape in 64 us.  A form that
        // also read the next stage's first fragments and issued the look-ahead loads under the MFMAs measured the same (76.6 / 47.1 us) and was not kept.
        // vmcnt bookkeeping: a wave issues, per stage, FW DMA instructions + KST * PT activation loads, all VMEM loads, which return in order — "at most
        // (D - 1) stage-issues outstanding" therefore means stage st's pieces (and its activations) have landed; the barrier behind that wait publishes
        // everybody's pieces and, since every wave is then past the multiplies of stage st - 1, frees slot (st - 1) % R = (st + D) % R for the next DMA.
        extern __shared__ __attribute__((aligned(16))) unsigned char lb_raw[];
        constexpr int KST = 2, R = 4, D = 3;
        constexpr int FR = KST * CT, FW = FR / 4;                         // fragments (KiB) per stage, per wave
        static_assert(FR % 4 == 0 && R * FR * 1024 <= 65536, "stage splits over the four waves; the ring stays inside one DS offset range");
        // the DMA as a BUFFER load (descriptor = this channel tile's fragments, per-lane part lane * 16, fragment offset in the scalar operand): the compiler files
        // global_load_lds under "flat access that may touch LDS and memory" and from then on waits for vmcnt(0) in front of every use of a loaded register —
        // the activation fragments of the stage three ahead included; a buffer load it counts like any other
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(w_all)) + ((size_t)(n_tile * CT) * w_steps) * 1024, 0, 0x7fffffff, 0x00020000);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);         // provably uniform: the fragment offset stays a scalar
        frag_t bst[R][KST][PT];
        // The activation loads of this branch: the addressing of load_b with everything a lane needs held in REGISTERS — the general loader re-reads kernel
        // arguments with scalar loads inside divergent branches (3x3: Hin, stride, channel offset per tap and pixel tile) and keeps the per-source offsets of a
        // concat in scratch; either stalls the issue phase of a stage for longer than its multiplies take.
        auto pin = [](auto v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };   // a kernel argument as a live scalar register, not a re-load
        const int kst_ = pin(a.ksteps), Cin_ = pin(a.Cin);
        const int str0 = pin(a.srcStride[0]), cof0 = pin(a.srcCoff[0]);
        int pixm[PT], pixu[PT];                                           // VAR_MULTI: linear pixel, pixel of the half-resolution grid
        unsigned int tapok[PT];                                           // VAR_3X3S2: bit tap = the tap is inside the image
        long long tap0[PT];                                               // VAR_3X3S2: element offset of tap (0, 0) (may be negative: never used when the tap is outside)
        int st1 = 0, st2 = 0, st3 = 0, co1 = 0, co2 = 0, co3 = 0, c1 = 0, c2 = 0, c3 = 0, sc0 = 0, sc1 = 0, sc2 = 0, sc3 = 0;
        bool up0 = false, up1 = false, up2 = false, up3 = false;
        const T* p1 = s0; const T* p2 = s0; const T* p3 = s0;
        if constexpr (VAR == VAR_MULTI) {
            st1 = pin(a.srcStride[1]); st2 = pin(a.srcStride[2]); st3 = pin(a.srcStride[3]);
            co1 = pin(a.srcCoff[1]); co2 = pin(a.srcCoff[2]); co3 = pin(a.srcCoff[3]);
            c1 = pin(a.cum[1]); c2 = pin(a.cum[2]); c3 = pin(a.cum[3]);
            sc0 = pin(a.srcC[0]); sc1 = pin(a.srcC[1]); sc2 = pin(a.srcC[2]); sc3 = pin(a.srcC[3]);
            up0 = a.srcMode[0] == MAF_SRC_UP2; up1 = a.nsrc > 1 && a.srcMode[1] == MAF_SRC_UP2; up2 = a.nsrc > 2 && a.srcMode[2] == MAF_SRC_UP2; up3 = a.nsrc > 3 && a.srcMode[3] == MAF_SRC_UP2;
            p1 = static_cast<const T*>(a.nsrc > 1 ? a.src[1] : a.src[0]); p2 = static_cast<const T*>(a.nsrc > 2 ? a.src[2] : a.src[0]); p3 = static_cast<const T*>(a.nsrc > 3 ? a.src[3] : a.src[0]);   // (constant indices: a computed one sends the argument block through scratch)
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = m_base + pt * 16 + p;
            const int mm = m < a.M ? m : 0;
            pixm[pt] = mm; pixu[pt] = mm; tapok[pt] = 0; tap0[pt] = 0;
            if constexpr (VAR != VAR_DIRECT) {
                const int x = mm % a.W, t = mm / a.W, y = t % a.H, b = t / a.H;
                if constexpr (VAR == VAR_MULTI) pixu[pt] = (b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
                if constexpr (VAR == VAR_3X3S2) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = 2 * y - 1 + tap / 3, ix = 2 * x - 1 + tap % 3;
                        if ((unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win) tapok[pt] |= 1u << tap;
                    }
                    tap0[pt] = ((long long)(b * a.Hin + 2 * y - 1) * a.Win + (2 * x - 1)) * str0 + cof0;
                }
            }
        }
        const int Win_ = VAR == VAR_3X3S2 ? pin(a.Win) : 0;
        auto load_fast = [&](int step, frag_t (&bf)[PT]) __attribute__((always_inline)) {
            const bool pad = step > last_step;                             // uniform
            const int sc = pad ? last_step : step;
            if (MAF_KO & 2) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) bf[pt] = (frag_t)(T)(0.001f * (float)(lane + step));
                return;
            }
            if constexpr (VAR == VAR_DIRECT) {
                int c0 = sc * F::KS + g * CH;
                c0 = c0 < Cin_ ? c0 : 0;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) bf[pt] = ldg16<T>(pad ? zpage : s0 + (size_t)pixm[pt] * str0 + cof0 + c0);
            } else if constexpr (VAR == VAR_3X3S2) {
                const int tap = sc / kst_, ks = sc - tap * kst_;
                const int ky = tap / 3, kx = tap - ky * 3;
                int c0 = ks * F::KS + g * CH;
                c0 = c0 < Cin_ ? c0 : 0;
                const long long toff = (long long)(ky * Win_ + kx) * str0 + c0;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const bool ok = !pad && ((tapok[pt] >> tap) & 1u);
                    const T* q = s0 + (tap0[pt] + toff);
                    bf[pt] = ldg16<T>(ok ? q : zpage);
                }
            } else {
                // every captured scalar read HERE, in front of the conditionals: read inside their arms, the loads (from the closure, before it is inlined away) are
                // merged into one load of a selected ADDRESS, and closure and variables stay in scratch for good (360 bytes, a waterfall loop around every DMA)
                const int c1v = c1, c2v = c2, c3v = c3, sc0v = sc0, sc1v = sc1, sc2v = sc2, sc3v = sc3, st0v = str0, st1v = st1, st2v = st2, st3v = st3;
                const int co0v = cof0, co1v = co1, co2v = co2, co3v = co3;
                const bool up0v = up0, up1v = up1, up2v = up2, up3v = up3;
                const T* const p0v = s0; const T* const p1v = p1; const T* const p2v = p2; const T* const p3v = p3;
                const bool i1 = sc >= c1v, i2 = sc >= c2v, i3 = sc >= c3v;  // (scalar) which source this k-step belongs to
                const int first = i3 ? c3v : i2 ? c2v : i1 ? c1v : 0;
                const int srcC = i3 ? sc3v : i2 ? sc2v : i1 ? sc1v : sc0v;
                const int str = i3 ? st3v : i2 ? st2v : i1 ? st1v : st0v;
                const int cof = i3 ? co3v : i2 ? co2v : i1 ? co1v : co0v;
                const bool up = i3 ? up3v : i2 ? up2v : i1 ? up1v : up0v;
                const T* sp = i3 ? p3v : i2 ? p2v : i1 ? p1v : p0v;
                int c0 = (sc - first) * F::KS + g * CH;
                c0 = c0 < srcC ? c0 : 0;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int pu = pixu[pt], pm = pixm[pt];
                    const int pix = up ? pu : pm;
                    bf[pt] = ldg16<T>(pad ? zpage : sp + (size_t)pix * str + cof + c0);
                }
            }
        };
        auto issue = [&](int st, auto slot_tag) __attribute__((always_inline)) {      // stage st -> ring slot (static)
            constexpr int slot = decltype(slot_tag)::value;
#pragma unroll
            for (int i = 0; i < FW; ++i) {
                const int f = wave_u * FW + i;                            // fragment of the stage: k-step u = f / CT, channel tile ct = f % CT
                const int u = f / CT, ct = f - u * CT;
                int step = st * KST + u;
                step = step > last_step ? last_step : step;               // padding steps re-read the last one (times zero activations)
                if (!(MAF_KO & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (void __attribute__((address_space(3)))*)(lb_raw + (slot * FR + f) * 1024), 16, lane * 16,
                                                         (ct * w_steps + step) * 1024, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < KST; ++u) load_fast(st * KST + u, bst[slot][u]);
        };
        const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(lb_raw + lane * 16);
        const int nstage = (total_steps + KST - 1) / KST;
        cm_static_for<D>([&](auto j) __attribute__((always_inline)) { issue(decltype(j)::value, j); });
        for (int st0 = 0; st0 < nstage; st0 += R) {
            cm_static_for<R>([&](auto q_tag) __attribute__((always_inline)) {
                constexpr int q = decltype(q_tag)::value;
                const int st = st0 + q;
                __builtin_amdgcn_sched_barrier(0);
                cm_wait_vm<(D - 1) * (FW + KST * PT)>();
                __builtin_amdgcn_s_barrier();                             // the bare instruction: __syncthreads() carries a fence that waits for vmcnt(0), i.e. for the whole look-ahead
                __builtin_amdgcn_sched_barrier(0);
                issue(st + D, std::integral_constant<int, (q + D) % R>{});
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the counter now counts only the reads below
                u32x4_t wr[FR];
                cm_static_for<FR>([&](auto f_tag) __attribute__((always_inline)) {
                    constexpr int f = decltype(f_tag)::value;
                    cm_ds_read_b128<(q * FR + f) * 1024>(wr[f], lbase);
                });
                cm_static_for<FR>([&](auto f_tag) __attribute__((always_inline)) {
                    constexpr int f = decltype(f_tag)::value, u = f / CT, ct = f % CT;
                    cm_wait_lgkm<FR - 1 - f>(wr[f]);
                    const frag_t wf = __builtin_bit_cast(frag_t, wr[f]);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        if (MAF_KO & 8) acc[pt][ct][0] += (float)bst[q][u][pt][0] + (float)wf[0];
                        else acc[pt][ct] = F::mma(bst[q][u][pt], wf, acc[pt][ct]);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the look-ahead stages past the end (clamped re-reads) are still landing
    } else if constexpr (LB) {
        // Long reductions (3x3 taps, wide concats): the four waves of a workgroup use the SAME weight fragments, and at
        // CT KiB per k-step per wave the L2 -> L1 weight stream, not HBM, bounds the layer.  Here the workgroup fetches each
        // k-step's CT fragments once (global -> registers during the previous step's MFMAs -> LDS, double-buffered, one
        // barrier per step) and every wave reads them with ds_read_b128; activations keep a private, deep register ring.
        extern __shared__ __attribute__((aligned(16))) unsigned char lb_raw[];
        frag_t* lb = reinterpret_cast<frag_t*>(lb_raw);                   // [3][CT * 64]: THREE buffers (see the loop)
        constexpr int NV = (CT * 64 + 255) / 256;
        const frag_t* wsrc = reinterpret_cast<const frag_t*>(w_all) + ((size_t)(n_tile * CT) * total_steps) * 64;
        constexpr int WD = 3;                                             // weight fragments leave global memory WD k-steps before their MFMAs
        frag_t wreg[WD][NV];
        auto gload = [&](int step, frag_t (&wr)[NV]) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + v * 256;
                if (MAF_KO & 1) { wr[v] = (frag_t)(T)(0.002f * (float)(lane + v)); continue; }
                if (idx < CT * 64) wr[v] = wsrc[((size_t)(idx >> 6) * total_steps + step) * 64 + (idx & 63)];
            }
        };
        auto lstore = [&](int buf, const frag_t (&wr)[NV]) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + v * 256;
                if (idx < CT * 64) lb[buf * (CT * 64) + idx] = wr[v];
            }
        };
        constexpr int NS = 6;                                             // activation loads run NS - 1 k-steps ahead (few, long wave-chains must keep many bytes in flight)
        static_assert(NS % WD == 0, "ring indices are static");
        frag_t bst[NS][PT];
        // Pipeline of a k-step s (one barrier per step):  weights(s + 2) registers -> LDS buffer (s + 2) % 3   |   fragments of step s + 1 read
        // from buffer (s + 1) % 3 into registers (published by the previous barrier)   |   MFMAs of step s on fragments that are ALREADY in
        // registers.  With two buffers the fragment read sat between the barrier and the MFMAs that need it: half of the MFMAs of the
        // <1,4,*> instantiations (the 20x20 / 40x40 layers: 22 launches per forward) waited a full LDS round trip (`s_waitcnt lgkmcnt(0)`).
#pragma unroll
        for (int s = 0; s < WD; ++s) gload(min(s, last_step), wreg[s]);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) load_b(s, bst[s]);
        lstore(0, wreg[0]);
        lstore(1, wreg[1]);
        gload(min(WD, last_step), wreg[0]);
        gload(min(WD + 1, last_step), wreg[1]);
        __syncthreads();
        frag_t wf[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wf[0][ct] = lb[ct * 64 + lane];                       // fragments of step 0
        // the trip count is rounded up to whole rings: padding steps multiply zero activations (zero page) with re-read weights,
        // so the loop body has no branch and every s_waitcnt counts exactly the loads that may stay in flight
        for (int step0 = 0; step0 < total_steps; step0 += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int step = step0 + u;
                load_b(step + NS - 1, bst[(u + NS - 1) % NS]);
                const frag_t* wl = lb + ((step + 1) % 3) * (CT * 64) + lane;                 // next step's fragments: in flight during this step's MFMAs
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wf[(u + 1) & 1][ct] = wl[ct * 64];
                mma_stage(bst[u], wf[u & 1]);
                lstore((step + 2) % 3, wreg[(u + 2) % WD]);                                   // weights of step + 2 (their buffer was last read two steps ago)
                gload(min(step + 2 + WD, last_step), wreg[(u + 2) % WD]);
                __syncthreads();
            }
        }
    } else {
    // Software pipeline: a ring of NS register stages, loads issued NS-1 k-steps ahead of their MFMAs.
    // Small-M layers (PT == 1: the 20x20 / 40x40 maps) are latency-bound chains of short k-steps, so they
    // run 4 stages deep; big layers (PT == 2) keep 2 stages and spend the registers on occupancy instead.
    // The trip count is rounded up to whole rings (padding steps: zero activations x re-read weights), so the loop body is
    // branch-free and the compiler's s_waitcnt lets exactly the prefetched stages stay in flight across the MFMAs.
    constexpr int NS = PT == 1 ? 4 : 2;
    frag_t bst[NS][PT], ast[NS][CT];
    if (s_end - s_begin == 1) {                                           // single k-step (Cin <= 32): nothing to pipeline
        load_b(s_begin, bst[0]);
        load_a(s_begin, ast[0]);
        mma_stage(bst[0], ast[0]);
    } else {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) {
            load_b(s_begin + s, bst[s]);
            load_a(s_begin + s, ast[s]);
        }
        for (int step0 = s_begin; step0 < s_end; step0 += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                load_b(step0 + u + NS - 1, bst[(u + NS - 1) % NS]);
                load_a(step0 + u + NS - 1, ast[(u + NS - 1) % NS]);
                mma_stage(bst[u], ast[u]);
            }
        }
    }
    }

    if constexpr (KS4) {      // reduce the four partial accumulators through LDS; wave 0 finishes the tile
        __shared__ f32x4_t red[3][CT][64];
        if (wave > 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) red[wave - 1][ct][lane] = acc[0][ct];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[0][ct] += red[0][ct][lane] + red[1][ct][lane] + red[2][ct][lane];
    }

    // ---- epilogue: bias + activation; lane (g, p) owns channels [cl, cl + CT) of pixels g*4 .. g*4+3 of each tile
    const int cl = n_tile * (16 * CT) + p * CT;
    float bias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bias[ct] = bias_all[cl + ct];          // padded to nN*16*CT on the host
    const int nvalid = a.Cout - cl;                                        // >= CT for every tile but the last

    // the activation is picked ONCE (a scalar branch around three straight-line copies of the epilogue), not per value: with the switch inside
    // the loops every value was its own basic block and every v_exp / v_rcp latency was exposed
    auto epilogue = [&](auto act_tag) {
        constexpr int ACT_ = decltype(act_tag)::value;
    #pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m_base + pt * 16 + g * 4 + r;
                if (m >= (dgs ? a.dg_mc : a.M)) continue;
                if ((MAF_KO & 4) && acc[pt][0][r] != 12345.678f) continue;
                if (dgs) {                                                    // class pixel -> linear pixel of the full-resolution output
                    const int w2 = a.W >> 1, h2 = a.H >> 1;
                    const int t = m / w2;
                    m = ((t / h2) * a.H + 2 * (t % h2) + dg_py) * a.W + 2 * (m - t * w2) + dg_px;
                }
                float v[CT];
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) v[ct] = ACT_ < 0 ? maf_act_rt(acc[pt][ct][r] + bias[ct], a.act) : maf_act<(ACT_ < 0 ? 0 : ACT_)>(acc[pt][ct][r] + bias[ct]);
                const size_t o = (size_t)m * a.out_stride + a.out_coff + cl;
                if (OUTF32 || sizeof(T) == 4) {
                    float* op = static_cast<float*>(out_all) + o;
                    if (nvalid >= CT) {
    #pragma unroll
                        for (int q = 0; q + 4 <= CT; q += 4) *reinterpret_cast<f32x4_t*>(op + q) = (f32x4_t){v[q], v[q + 1], v[q + 2], v[q + 3]};
                        if (CT % 4 == 2) *reinterpret_cast<float2*>(op + CT - 2) = make_float2(v[CT - 2], v[CT - 1]);
                    } else {
    #pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (ct < nvalid) op[ct] = v[ct];
                    }
                } else {
                    half_t* op = static_cast<half_t*>(out_all) + o;
                    if (nvalid >= CT) {
                        uint32_t w[CT / 2];
    #pragma unroll
                        for (int q = 0; q < CT / 2; ++q) {
                            const half2_t h = {(half_t)v[2 * q], (half_t)v[2 * q + 1]};
                            w[q] = __builtin_bit_cast(uint32_t, h);
                        }
                        if (CT == 8) *reinterpret_cast<u32x4_t*>(op) = (u32x4_t){w[0], w[1], w[2], w[3 % (CT / 2)]};
                        else if (CT == 6) { *reinterpret_cast<u32x2_t*>(op) = (u32x2_t){w[0], w[1]}; *reinterpret_cast<uint32_t*>(op + 4) = w[2 % (CT / 2)]; }
                        else if (CT == 4) *reinterpret_cast<u32x2_t*>(op) = (u32x2_t){w[0], w[1 % (CT / 2)]};
                        else *reinterpret_cast<uint32_t*>(op) = w[0];
                    } else {
    #pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            if (ct < nvalid) op[ct] = (half_t)v[ct];
                    }
                }
            }
        }
    };
    if (a.act == MAF_ACT_SILU) epilogue(std::integral_constant<int, MAF_ACT_SILU>());
    else if (a.act == MAF_ACT_RELU) epilogue(std::integral_constant<int, MAF_ACT_RELU>());
    else if (a.act == MAF_ACT_NONE) epilogue(std::integral_constant<int, MAF_ACT_NONE>());
    else epilogue(std::integral_constant<int, -1>());
}

template <typename T, int PT, int CT, int VAR, bool OUTF32, bool KS4 = false>
int launch_act(const ConvArgs& a, hipStream_t s) {
    const int grid = maf_cdiv(a.nM, 8) * 8 * a.nN;
    hipLaunchKernelGGL((conv_mfma_kernel<T, PT, CT, VAR, OUTF32, KS4>), dim3(grid, a.twin ? 2 : 1), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "conv_mfma launch");
}

template <typename T, int VAR, bool OUTF32>
int launch_tile(const ConvArgs& a, int pt, int ct, hipStream_t s) {
#define MAF_TILE(P, C) \
    if (pt == P && ct == C) return launch_act<T, P, C, VAR, OUTF32>(a, s);
    MAF_TILE(1, 2) MAF_TILE(2, 2) MAF_TILE(4, 2) MAF_TILE(1, 4) MAF_TILE(2, 4) MAF_TILE(4, 4)
    MAF_TILE(1, 6) MAF_TILE(2, 6) MAF_TILE(1, 8) MAF_TILE(2, 8)
#undef MAF_TILE
    if (pt == 0) {                                                   // tile_p == 0 selects the split-K variant (tile_k = 4)
        if (ct == 2) return launch_act<T, 1, 2, VAR, OUTF32, true>(a, s);
        if (ct == 4) return launch_act<T, 1, 4, VAR, OUTF32, true>(a, s);
        if (ct == 6) return launch_act<T, 1, 6, VAR, OUTF32, true>(a, s);
        if (ct == 8) return launch_act<T, 1, 8, VAR, OUTF32, true>(a, s);
    }
    maf_set_error("conv: unsupported tile (tile_p in {1,2,4}, tile_c in {2,4,6,8}; tile_p = 4 only with tile_c <= 4)");
    return MAF_E_UNSUPPORTED;
}

template <int PT, int CT, int VAR>
int launch_lb(const ConvArgs& a, hipStream_t s) {
    const int grid = maf_cdiv(a.nM, 8) * 8 * a.nN;
    hipLaunchKernelGGL((conv_mfma_kernel<half_t, PT, CT, VAR, false, false, true>), dim3(grid, a.twin ? 2 : 1), dim3(256), 3 * CT * 1024, s, a);
    return maf_check_hip(hipGetLastError(), "conv_mfma (LDS-shared weights) launch");
}

template <int VAR>
int launch_lb_tile(const ConvArgs& a, int pt, int ct, hipStream_t s) {
#define MAF_LB(P, C) \
    if (pt == P && ct == C) return launch_lb<P, C, VAR>(a, s);
    MAF_LB(1, 4) MAF_LB(2, 4) MAF_LB(4, 4) MAF_LB(1, 6) MAF_LB(2, 6) MAF_LB(1, 8) MAF_LB(2, 8)
#undef MAF_LB
    maf_set_error("conv: tile_k = 2 (LDS-shared weights) supports tile_c in {4,6,8}, tile_p in {1,2} (4 with tile_c 4)");
    return MAF_E_UNSUPPORTED;
}

template <int PT, int CT, int VAR>
int launch_dma(const ConvArgs& a, hipStream_t s) {
    const int grid = maf_cdiv(a.nM, 8) * 8 * a.nN;
    hipLaunchKernelGGL((conv_mfma_kernel<half_t, PT, CT, VAR, false, false, true, true>), dim3(grid, a.twin ? 2 : 1), dim3(256), 4 * 2 * CT * 1024, s, a);
    return maf_check_hip(hipGetLastError(), "conv_mfma (weights by DMA ring) launch");
}

template <int VAR>
int launch_dma_tile(const ConvArgs& a, int pt, int ct, hipStream_t s) {
#define MAF_DMA(P, C) \
    if (pt == P && ct == C) return launch_dma<P, C, VAR>(a, s);
    MAF_DMA(1, 4) MAF_DMA(2, 4) MAF_DMA(1, 6) MAF_DMA(2, 6) MAF_DMA(1, 8) MAF_DMA(2, 8)
#undef MAF_DMA
    maf_set_error("conv: tile_k = 8 (weights by DMA ring) supports tile_c in {4,6,8}, tile_p in {1,2}");
    return MAF_E_UNSUPPORTED;
}

template <typename T>
int launch_var(const ConvArgs& a, int var, bool outf32, int pt, int ct, hipStream_t s) {
    if (var == VAR_DIRECT) return outf32 ? launch_tile<T, VAR_DIRECT, true>(a, pt, ct, s) : launch_tile<T, VAR_DIRECT, false>(a, pt, ct, s);
    if (var == VAR_MULTI) return launch_tile<T, VAR_MULTI, false>(a, pt, ct, s);
    if (var == VAR_POOL2) return launch_tile<T, VAR_POOL2, false>(a, pt, ct, s);
    return launch_tile<T, VAR_3X3S2, false>(a, pt, ct, s);
}

}  // namespace

