// MAF_OP_HEADTAIL: the tail of one detection level in ONE launch — for both branches of Head_DepthUni (yolov6/layers/common.py:1288-1336)
//   cls:  cls_conv_s (1x1 + SiLU)  ->  cls_pred (1x1)  ->  sigmoid                    -> columns 5.. of the prediction rows
//   reg:  reg_conv_s (1x1 + SiLU)  ->  reg_pred (1x1)  ->  DFL softmax expectation, dist2bbox('xywh'), x stride, objectness 1
//                                                                                       -> columns 0..4
// i.e. Conv.forward_fuse x2 + the level's share of the Detect_yaml eval branch (yolov6/models/yolo.py:355-396).  Replaces four
// MAF_OP_CONV1X1 launches per level and the level's part of MAF_OP_DECODE; the 128-channel mid activations and the fp32 logits never
// reach HBM (per anchor: 2*C fp16 read + 85 fp32 written, instead of 2*C + 4*C + 2*148*4 + 85*4 bytes).
//
// Both GEMMs run on the matrix cores without a transposition in between: the first one is computed transposed (A = weight fragments,
// B = 16 pixels x 32 input channels = ONE 16-byte load per lane), so a lane ends up holding, for its pixel, channels 16t + 4G + r of every
// 16-channel tile t — which IS the A-operand layout (pixel row, 8 consecutive k-slots per lane) of the second GEMM once the second
// weight matrix is packed with its K axis in that order (maf-yolo_amd/pack.py:pack_head_tail).  Weights of a branch sit in LDS for the
// whole workgroup; the epilogue goes through a wave-private LDS tile so the prediction rows are written as full contiguous runs.
// Head widths 256 and 384 (P5 of s, every level of m): a branch's weights (172 / 356 KB) do not fit the LDS, so those instantiations
// (WLDS = false) read the same fragment records straight from global memory — every wave streams them once per 16-pixel unit out of the
// L2, where they stay resident (the records of a level total <= 0.7 MB).
#include "maf_common.h"
#include "lds_pipe.h"
#include <type_traits>

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko KO_SRCS=head_tail.hip): 1 = no activation loads, 2 = no first GEMM, 4 = no SiLU between the GEMMs, 8 = no second GEMM, 16 = no epilogue at all, 32 = no prediction stores
#endif

namespace {

template <int N, int I = 0, typename F>
__device__ __forceinline__ void ht_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ht_static_for<N, I + 1>(f);
    }
}

constexpr int NT2 = 5;                 // 80 output columns of the second GEMM (80 classes; 68 DFL logits + zero padding)
constexpr int NO = 85, NC = 80, NR = 68;

struct HtArgs {
    const half_t* x[2];                // [M][xs] fp16, first channel xc: input of cls_conv_s / reg_conv_s
    int xs[2], xc[2];
    const char* rec[2];                // weight records (pack_head_tail)
    float* out;                        // pred [B][A][85]
    int M, HW, Wd, A, lvl_off, iters;  // pixels of the level (B*H*W), H*W, grid width, anchors per image, first anchor of the level
    float stride;
    // fused candidate filter (op->aux[1] != 0): the class scores this kernel has just computed are tested against non_max_suppression's conf_thres
    // and the survivors appended to the NMS workspace — what nms_collect_multi_kernel (csrc/nms.hip) otherwise re-reads the whole prediction for
    int* cand_cnt;                     // [B][MAF_NMS_CNT_STRIDE] candidate counters of the NMS workspace
    unsigned long long* cand_keys;     // [B][cand_cap] 64-bit keys (~score bits << 32 | box * 80 + class)
    long long cand_cap;
    float cand_conf;
};

template <int C, int PT, bool FILTER = false>      // FILTER: also append the NMS candidates (a separate instantiation: the plain one pays nothing for it)
__global__ __launch_bounds__(256, C <= 128 ? 2 : 1) void head_tail_kernel(const HtArgs a) {
    constexpr int KS = C / 32, T1 = C / 16;
    constexpr int W1B = C * C * 2, W2B = 80 * C * 2;
    constexpr bool WLDS = C <= 192;                                               // the record fits the LDS
    constexpr int CHB = KS * 1024, CHV = (CHB / 16 + 255) / 256, NCH = T1 + NT2;     // chunked instantiations: bytes per chunk, 16-byte pieces per thread, chunks
    __shared__ __attribute__((aligned(16))) char s_w[WLDS ? W1B + W2B + C * 4 + 80 * 4 : 2 * CHB + C * 4 + 80 * 4];
    __shared__ __attribute__((aligned(16))) float s_stage[4][16 * NC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, n = lane & 15;
    const int br = blockIdx.y;
    if constexpr (WLDS) {
        // the branch's record travels global -> LDS by DMA (1 KiB per wave-instruction, all pieces in flight at once, no registers); the
        // barrier behind the first activation loads publishes it
        constexpr int NB = W1B + W2B + C * 4 + 80 * 4;
        static_assert(NB % 16 == 0, "record in 16-byte pieces");
        const char* src = a.rec[br];
        for (int v0 = wave * 64; v0 < NB / 16; v0 += 256)
            if (v0 + lane < NB / 16)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)(v0 + lane) * 16),
                                                 (void __attribute__((address_space(3)))*)(s_w + v0 * 16), 16, 0, 0);
    }
    // chunked instantiations: the fragment stream of the record (W1 tiles, then W2 tiles: NCH chunks of CHB bytes, contiguous) goes through
    // s_w[0 .. 2 CHB); the biases sit behind it
    const uint4* const rec16 = reinterpret_cast<const uint4*>(a.rec[br]);
    auto chunk_load = [&](int c, uint4 (&r)[CHV]) {
#pragma unroll
        for (int v = 0; v < CHV; ++v) {
            const int i = tid + v * 256;
            if (CHB / 16 % 256 == 0 || i < CHB / 16) r[v] = rec16[(size_t)c * (CHB / 16) + i];
        }
    };
    auto chunk_store = [&](int buf, const uint4 (&r)[CHV]) {
#pragma unroll
        for (int v = 0; v < CHV; ++v) {
            const int i = tid + v * 256;
            if (CHB / 16 % 256 == 0 || i < CHB / 16) reinterpret_cast<uint4*>(s_w + buf * CHB)[i] = r[v];
        }
    };
    if constexpr (!WLDS) {
        uint4 r0[CHV];
        chunk_load(0, r0);
        chunk_store(0, r0);
        const uint4* bsrc = reinterpret_cast<const uint4*>(a.rec[br] + W1B + W2B);
        for (int i = tid; i < (C * 4 + 80 * 4) / 16; i += 256) reinterpret_cast<uint4*>(s_w + 2 * CHB)[i] = bsrc[i];
        __syncthreads();
    }
    constexpr int BOFF = WLDS ? W1B + W2B : 2 * CHB;
    const half8_t* w1 = reinterpret_cast<const half8_t*>(s_w);
    const half8_t* w2 = reinterpret_cast<const half8_t*>(s_w + W1B);
    uint32_t w1a[2], w2a[2];                                                      // LDS addresses of this lane's 16 bytes of fragment 0 / 64 of W1 and W2
#pragma unroll
    for (int k = 0; k < 2; ++k) { w1a[k] = lp_lds_addr(s_w + k * 65536 + lane * 16); w2a[k] = lp_lds_addr(s_w + W1B + k * 65536 + lane * 16); }
    const f32x4_t* b1 = reinterpret_cast<const f32x4_t*>(s_w + BOFF);
    const float* b2 = reinterpret_cast<const float*>(s_w + BOFF + C * 4);
    const half_t* xb = a.x[br] + a.xc[br] + 8 * G;
    const int xs = a.xs[br];
    float* stage = s_stage[wave];

    half8_t x[PT][KS];
    auto load_x = [&](int unit) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = min((unit * PT + p) * 16 + n, a.M - 1);                // clamped: loads stay unconditional
            const half_t* px = xb + (size_t)m * xs;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { if (MAF_KO & 1) x[p][ks] = (half8_t)(half_t)0.01f; else x[p][ks] = *reinterpret_cast<const half8_t*>(px + 32 * ks); }
        }
    };
    // persistent: the workgroups of a branch walk the 16 * PT-pixel units with the stride of the grid (a.iters rounds, the same for every
    // workgroup: the epilogue's barriers need uniform trip counts; units past the end are computed on clamped pixels and never stored)
    const int ustride = gridDim.x * 4;
    const int unit0 = blockIdx.x * 4 + wave;
    load_x(unit0);
    if constexpr (WLDS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float bias2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) bias2[t] = b2[16 * t + n];
    for (int it = 0; it < a.iters; ++it) {
        const int unit = unit0 + it * ustride;
        const int par = WLDS ? 0 : (it * NCH) & 1;                                // chunked: chunk c of this unit sits in buffer (c + par) & 1 (NCH is odd)
        // ---- GEMM 1 (transposed): acc1[p][t] lane (G, n) = channels 16t + 4G + r of pixel n
        constexpr int T1R = WLDS ? T1 : 1;                                        // chunked instantiations turn every pair of tiles into its fragment at once
        f32x4_t acc1[PT][T1R];
        half8_t a2[PT][KS];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int t = 0; t < T1R; ++t) acc1[p][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (WLDS) {
            // T1 * KS steps, one straight line, the weight fragment of step s + RD read before the MFMAs of step s (lds_pipe.h)
            constexpr int NSTEP = T1 * KS, RD = 6;
            u32x4_t wr[RD + 1];
            auto ld_step = [&](auto idx) {
                constexpr int s_ = decltype(idx)::value;
                if constexpr (s_ < NSTEP) lp_ds_read_b128<(s_ * 1024) % 65536>(wr[s_ % (RD + 1)], w1a[(s_ * 1024) / 65536]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lp_static_for<RD>([&](auto idx) { ld_step(idx); });
            lp_static_for<NSTEP>([&](auto idx) {
                constexpr int s_ = decltype(idx)::value, t = s_ / KS, ks = s_ % KS, sl = s_ % (RD + 1);
                ld_step(std::integral_constant<int, s_ + RD>{});
                constexpr int ahead = (NSTEP - 1 - s_) < RD ? (NSTEP - 1 - s_) : RD;
                lp_wait_lgkm<ahead>(wr[sl]);
                const half8_t wa = __builtin_bit_cast(half8_t, wr[sl]);
#pragma unroll
                for (int p = 0; p < PT; ++p) if (!(MAF_KO & 2)) acc1[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, x[p][ks], acc1[p][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            // the record does not fit the LDS: its T1 + NT2 chunks (the fragments of one 16-channel tile: KS KiB) pass through two LDS
            // buffers — chunk c + 1 travels global -> registers -> LDS while chunk c is multiplied, one barrier per chunk
            f32x4_t even[PT];
            ht_static_for<T1>([&](auto idx) {
                constexpr int t = decltype(idx)::value;
                uint4 nxt[CHV];
                chunk_load(t + 1, nxt);
                const half8_t* wc = reinterpret_cast<const half8_t*>(s_w + ((t + par) & 1) * CHB);
                f32x4_t acc[PT];
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8_t wa = wc[ks * 64 + lane];
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, x[p][ks], acc[p], 0, 0, 0);
                }
                if constexpr (t % 2 == 0) {
#pragma unroll
                    for (int p = 0; p < PT; ++p) even[p] = acc[p];
                } else {                                                           // tiles 2j, 2j + 1 -> A fragment j of GEMM 2 (same rounding points as below)
                    constexpr int j = t / 2;
                    const f32x4_t bl = b1[4 * (2 * j) + G], bh = b1[4 * (2 * j + 1) + G];
#pragma unroll
                    for (int p = 0; p < PT; ++p)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            a2[p][j][r] = (half_t)maf_act<MAF_ACT_SILU>(even[p][r] + bl[r]);
                            a2[p][j][4 + r] = (half_t)maf_act<MAF_ACT_SILU>(acc[p][r] + bh[r]);
                        }
                }
                chunk_store((t + 1 + par) & 1, nxt);
                __syncthreads();
            });
        }
        if (it + 1 < a.iters) load_x(unit + ustride);                             // next unit's activations fly during the rest
        // ---- bias + SiLU -> fp16: the A fragments of GEMM 2 (k-slots q < 4 from tile 2j, q >= 4 from tile 2j + 1)
        if constexpr (WLDS)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const f32x4_t bl = b1[4 * (2 * j) + G], bh = b1[4 * (2 * j + 1) + G];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a2[p][j][r] = (MAF_KO & 4) ? (half_t)(acc1[p][2 * j][r] + bl[r]) : (half_t)maf_act<MAF_ACT_SILU>(acc1[p][2 * j][r] + bl[r]);
                    a2[p][j][4 + r] = (MAF_KO & 4) ? (half_t)(acc1[p][2 * j + 1][r] + bh[r]) : (half_t)maf_act<MAF_ACT_SILU>(acc1[p][2 * j + 1][r] + bh[r]);
                }
            }
        }
        // ---- GEMM 2: acc2[p][t2] lane (G, n) = pixels 4G + r, output column 16 t2 + n
        f32x4_t acc2[PT][NT2];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int t = 0; t < NT2; ++t) acc2[p][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (WLDS) {
            constexpr int NSTEP = NT2 * KS, RD = 6;
            u32x4_t wr[RD + 1];
            auto ld_step = [&](auto idx) {
                constexpr int s_ = decltype(idx)::value;
                if constexpr (s_ < NSTEP) lp_ds_read_b128<(s_ * 1024) % 65536>(wr[s_ % (RD + 1)], w2a[(s_ * 1024) / 65536]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lp_static_for<RD>([&](auto idx) { ld_step(idx); });
            lp_static_for<NSTEP>([&](auto idx) {
                constexpr int s_ = decltype(idx)::value, t = s_ / KS, j = s_ % KS, sl = s_ % (RD + 1);
                ld_step(std::integral_constant<int, s_ + RD>{});
                constexpr int ahead = (NSTEP - 1 - s_) < RD ? (NSTEP - 1 - s_) : RD;
                lp_wait_lgkm<ahead>(wr[sl]);
                const half8_t wb = __builtin_bit_cast(half8_t, wr[sl]);
#pragma unroll
                for (int p = 0; p < PT; ++p) if (!(MAF_KO & 8)) acc2[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[p][j], wb, acc2[p][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            ht_static_for<NT2>([&](auto idx) {
                constexpr int t = decltype(idx)::value, c = T1 + t;
                uint4 nxt[CHV];
                chunk_load(c + 1 < NCH ? c + 1 : 0, nxt);                       // after the last chunk: chunk 0 again, for the next unit
                const half8_t* wc = reinterpret_cast<const half8_t*>(s_w + ((c + par) & 1) * CHB);
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const half8_t wb = wc[j * 64 + lane];
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc2[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[p][j], wb, acc2[p][t], 0, 0, 0);
                }
                chunk_store((c + 1 + par) & 1, nxt);
                __syncthreads();
            });
        }
        // ---- epilogue: through the wave's LDS tile to whole prediction rows
        if ((MAF_KO & 16) && acc2[0][0][0] != 12345.678f) continue;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m0 = (unit * PT + p) * 16;
            // the staging tile is PRIVATE to the wave (s_stage[wave]) and a wave's LDS operations execute in program order: nothing to wait for but
            // the compiler's own reordering — no workgroup barrier, so the four waves drift apart and one wave's epilogue runs under another's GEMMs
            // (the chunked instantiations keep the barriers: their chunk loop needs the waves in step)
            if constexpr (WLDS) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads();
            if (br == 0) {
#pragma unroll
                for (int t = 0; t < NT2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[(4 * G + r) * NC + 16 * t + n] = maf_act<MAF_ACT_SIGMOID>(acc2[p][t][r] + bias2[t]);
            } else {
#pragma unroll
                for (int t = 0; t < NT2; ++t)
                    if (16 * t + n < NR) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[(4 * G + r) * NR + 16 * t + n] = acc2[p][t][r] + bias2[t];
                    }
            }
            if constexpr (WLDS) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads();
            if (br == 0) {
                if (a.HW % 16 == 0 && !(MAF_KO & 64)) {
                    // the 16 pixels of a unit are 16 CONSECUTIVE rows of one image's prediction: row r of the unit = dwords [D0 + 85 r + 0, + 80) of the
                    // output.  Nothing in a 340-byte row is 16-byte aligned by itself, but h_r = (-(D0 + 85 r)) & 3 dwords into the row the address is: a row
                    // is <= 3 head dwords, 19 or 20 aligned 16-byte pieces and the rest — 5 wave-wide 16-byte stores + ONE wave-wide dword store per unit
                    // instead of 20 dword stores (knock-outs: the prediction stores were 21 of the kernel's 62 us at 80 x 80)
                    const int b = m0 / a.HW, pin0 = m0 - b * a.HW;
                    float* const o0 = a.out + ((size_t)b * a.A + a.lvl_off + pin0) * NO + 5;
                    const unsigned int d0 = (unsigned int)((reinterpret_cast<uintptr_t>(o0) >> 2) & 3);              // (whatever the alignment of the caller's tensor)
                    const int rows = min(16, a.M - m0);
                    if (!(MAF_KO & 32)) {
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            const int id = lane + 64 * q, r = id / 20, k = id - r * 20;
                            const int h = (int)((0u - (d0 + 85u * (unsigned int)r)) & 3u), col = h + 4 * k;
                            if (r < rows && col + 4 <= NC) {
                                const float* sp = stage + r * NC + col;
                                const float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                                *reinterpret_cast<float4*>(o0 + (size_t)r * NO + col) = v;
                            }
                        }
                        {
                            const int r = lane >> 2, j = lane & 3;
                            const int h = (int)((0u - (d0 + 85u * (unsigned int)r)) & 3u);
                            const int col = j < h ? j : h + 4 * ((NC - h) >> 2) + (j - h);
                            if (r < rows && h != 0 && col < NC) o0[(size_t)r * NO + col] = stage[r * NC + col];
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < 16 * NC / 64; ++q) {
                    const int e = lane + 64 * q;
                    const int px = e / NC, col = e - px * NC;
                    const int m = m0 + px;
                    if (m < a.M) {
                        const int b = m / a.HW, pin = m - b * a.HW;
                        if (!(MAF_KO & 32)) a.out[((size_t)b * a.A + a.lvl_off + pin) * NO + 5 + col] = stage[e];
                    }
                }
                }
                if constexpr (FILTER) {                                           // the candidate filter of the NMS call that follows: a second walk over the tile
                    unsigned int hits = 0;
                    int mine = 0;
#pragma unroll
                    for (int q = 0; q < 16 * NC / 64; ++q) {
                        const int e = lane + 64 * q;
                        const bool ok = m0 + e / NC < a.M && stage[e] > a.cand_conf;  // nms.py:48, :69, :76 with objectness 1 (written by the other branch): score = cls * 1
                        hits |= ok ? 1u << q : 0u;
                        mine += __popcll(__ballot(ok));                           // wave-uniform running count
                    }
                    if (mine > 0) {                                               // the 16 pixels of a unit belong to ONE image (HW % 16 == 0: checked by the launcher)
                        const int b = m0 / a.HW, pin0 = m0 - b * a.HW;
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&a.cand_cnt[b * MAF_NMS_CNT_STRIDE], mine);
                        int pos = __shfl(base, 0);
                        unsigned long long* keys = a.cand_keys + (size_t)b * a.cand_cap;
#pragma unroll
                        for (int q = 0; q < 16 * NC / 64; ++q) {
                            const bool ok = (hits >> q) & 1u;
                            const unsigned long long mk = __ballot(ok);
                            if (ok) {
                                const int e = lane + 64 * q;
                                const int px = e / NC, col = e - px * NC;
                                const unsigned int flat = (unsigned int)(a.lvl_off + pin0 + px) * (unsigned int)NC + (unsigned int)col;
                                keys[pos + __popcll(mk & ((1ull << lane) - 1ull))] = ((unsigned long long)(~__float_as_uint(stage[e])) << 32) | flat;
                            }
                            pos += __popcll(mk);
                        }
                    }
                }
            } else {
                const int px = lane >> 2, side = lane & 3;
                const float* rr = stage + px * NR + side * 17;                     // channel = side * 17 + bin (yolo.py:376)
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 17; ++i) mx = fmaxf(mx, rr[i]);
                float se = 0.f, sw = 0.f;
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    const float e = __expf(rr[i] - mx);
                    se += e;
                    sw += e * (float)i;
                }
                const float d = sw / se;                                          // expected ltrb distance in grid units
                const int l0 = lane & ~3;
                const float lft = __shfl(d, l0), top = __shfl(d, l0 + 1), rgt = __shfl(d, l0 + 2), bot = __shfl(d, l0 + 3);
                const int m = m0 + px;
                if (m < a.M) {
                    const int b = m / a.HW, pin = m - b * a.HW;
                    const float ax = (float)(pin % a.Wd) + 0.5f, ay = (float)(pin / a.Wd) + 0.5f;
                    const float x1 = ax - lft, y1 = ay - top, x2 = ax + rgt, y2 = ay + bot;
                    float v = side == 0 ? (x1 + x2) * 0.5f : side == 1 ? (y1 + y2) * 0.5f : side == 2 ? x2 - x1 : y2 - y1;
                    float* o = a.out + ((size_t)b * a.A + a.lvl_off + pin) * NO;
                    if (!(MAF_KO & 32)) {
                    o[side] = v * a.stride;
                    if (side == 0) o[4] = 1.0f;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int64_t maf_head_tail_record_bytes(int32_t C) { return (int64_t)C * C * 2 + 80ll * C * 2 + C * 4 + 80 * 4; }

int maf_launch_head_tail(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "head_tail: fp16 only");
    MAF_REQUIRE(op->Cin == 128 || op->Cin == 192 || op->Cin == 64 || op->Cin == 256 || op->Cin == 384, "head_tail: head width must be 64, 128, 192, 256 or 384");
    MAF_REQUIRE(op->nc == 80 && op->reg_max == 16, "head_tail: 80 classes and reg_max 16");
    MAF_REQUIRE(op->nsrc == 2 && op->src[0].ptr && op->src[1].ptr && op->w && op->aux[0] && op->out, "head_tail: null pointer");
    MAF_REQUIRE(op->B > 0 && op->H > 0 && op->W > 0 && (long long)op->B * op->H * op->W < (1ll << 31), "head_tail: bad dims");
    HtArgs a;
    for (int i = 0; i < 2; ++i) {
        const maf_src_t& sr = op->src[i];
        MAF_REQUIRE(sr.C == op->Cin && sr.stride % 8 == 0 && sr.coff % 8 == 0 && sr.mode == MAF_SRC_DIRECT, "head_tail: sources must be direct, 16-byte aligned, Cin wide");
        a.x[i] = static_cast<const half_t*>(sr.ptr); a.xs[i] = sr.stride; a.xc[i] = sr.coff;
    }
    a.rec[0] = static_cast<const char*>(op->w); a.rec[1] = static_cast<const char*>(op->aux[0]);
    a.out = static_cast<float*>(op->out);
    a.M = op->B * op->H * op->W; a.HW = op->H * op->W; a.Wd = op->W; a.A = op->Win; a.lvl_off = op->Hin; a.stride = op->lvl_stride[0];
    MAF_REQUIRE(a.A >= a.lvl_off + a.HW && a.lvl_off >= 0, "head_tail: Hin = first anchor of the level, Win = anchors per image");
    a.cand_cnt = static_cast<int*>(const_cast<void*>(op->aux[1]));
    a.cand_keys = static_cast<unsigned long long*>(const_cast<void*>(op->aux[2]));
    a.cand_cap = op->lvl_h[0]; a.cand_conf = op->lvl_stride[1];
    MAF_REQUIRE(!a.cand_cnt || (a.cand_keys && a.cand_cap >= (long long)a.A * NC && a.HW % 16 == 0 && a.cand_conf >= 0.f && a.cand_conf < 1.f),
                "head_tail: candidate filter (aux[1] = counters, aux[2] = keys, lvl_h[0] = keys per image >= A * 80, lvl_stride[1] = conf in [0, 1)) needs H * W a multiple of 16");
    const int pt = op->Cin <= 128 ? 2 : 1;
    const int units = maf_cdiv(maf_cdiv(a.M, 16), pt);
    // one round of resident workgroups per branch (tile_k > 0: that many workgroups per CU and branch-pair, i.e. grid.x = 128 * tile_k; default 2 per CU
    // for the LDS-resident widths <= 128, 1 above), every workgroup walks ceil(units / (4 * grid.x)) units per wave
    const int per_cu = op->tile_k > 0 ? op->tile_k : (op->Cin <= 128 ? 2 : 1);
    int gx = std::min(maf_cdiv(units, 4), std::max(1, 128 * per_cu));
    a.iters = maf_cdiv(units, 4 * gx);
    gx = maf_cdiv(units, 4 * a.iters);                                             // same rounds, no idle workgroups
    const dim3 grid(gx, 2);
    if (a.cand_cnt) {
        switch (op->Cin) {
            case 64: hipLaunchKernelGGL((head_tail_kernel<64, 2, true>), grid, dim3(256), 0, s, a); break;
            case 128: hipLaunchKernelGGL((head_tail_kernel<128, 2, true>), grid, dim3(256), 0, s, a); break;
            case 192: hipLaunchKernelGGL((head_tail_kernel<192, 1, true>), grid, dim3(256), 0, s, a); break;
            case 256: hipLaunchKernelGGL((head_tail_kernel<256, 1, true>), grid, dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((head_tail_kernel<384, 1, true>), grid, dim3(256), 0, s, a); break;
        }
        return maf_check_hip(hipGetLastError(), "head_tail launch");
    }
    switch (op->Cin) {
        case 64: hipLaunchKernelGGL((head_tail_kernel<64, 2>), grid, dim3(256), 0, s, a); break;
        case 128: hipLaunchKernelGGL((head_tail_kernel<128, 2>), grid, dim3(256), 0, s, a); break;
        case 192: hipLaunchKernelGGL((head_tail_kernel<192, 1>), grid, dim3(256), 0, s, a); break;
        case 256: hipLaunchKernelGGL((head_tail_kernel<256, 1>), grid, dim3(256), 0, s, a); break;      // weights streamed from L2 (no LDS copy)
        default: hipLaunchKernelGGL((head_tail_kernel<384, 1>), grid, dim3(256), 0, s, a); break;
    }
    return maf_check_hip(hipGetLastError(), "head_tail launch");
}
