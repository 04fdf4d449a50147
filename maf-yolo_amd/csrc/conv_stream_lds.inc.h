// Kernel + launcher template of the persistent LDS-resident-weight 1x1 conv (instantiated by conv_stream_lds.hip for ksteps <= 12 and by
// conv_stream_lds_wide.hip for 13 .. 24): see conv_stream_lds.hip for the description.
#pragma once
#include "conv_mfma.inc.h"

namespace {

// LDS reads of the weight fragments as inline assembly with hand-counted waits: as plain loads the compiler reads two fragments into the same two
// register quads right in front of the two MFMAs that need them and waits with lgkmcnt(1) / lgkmcnt(0) — a full LDS round trip (~130 cycles) per
// pair of 17-cycle MFMAs, 4.7k cycles for the 72 MFMAs of a 288 -> 128 tile instead of 1.2k.
template <int OFF> __device__ __forceinline__ void sl_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void sl_wait_lgkm(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }   // "+v": the MFMA that reads `a` cannot move above the wait

template <int N, int I = 0, typename F>
__device__ __forceinline__ void sl_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sl_static_for<N, I + 1>(f);
    }
}

template <int ACT, int CT>
__device__ __forceinline__ void sl_store(const f32x4_t (&acc)[CT], int r, half_t* op, int nvalid) {
    float v[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) v[ct] = (MAF_KO & 16) ? acc[ct][r] : maf_act<ACT>(acc[ct][r]);
    if (nvalid >= CT) {
        uint32_t w[CT / 2];
#pragma unroll
        for (int c2 = 0; c2 < CT / 2; ++c2) {
            const half2_t h = {(half_t)v[2 * c2], (half_t)v[2 * c2 + 1]};
            w[c2] = __builtin_bit_cast(uint32_t, h);
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

// The same values as a PIXEL PAIR (MAF_SRC_PAIRS, ConvArgs.out_pairs): rows r0 and r0 + 1 of the accumulators are pixels 2q and 2q + 1; channel c of
// the pair is the dword (y[2q][c], y[2q+1][c]) at half 2 (coff + c) of the pair's 2 x stride block — CT dwords, 16-byte stores where CT allows.
template <int ACT, int CT>
__device__ __forceinline__ void sl_store_pair(const f32x4_t (&acc)[CT], int r0, half_t* op, int nvalid, bool second) {
    uint32_t w[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float lo = (MAF_KO & 16) ? acc[ct][r0] : maf_act<ACT>(acc[ct][r0]);
        const float hi = second ? ((MAF_KO & 16) ? acc[ct][r0 + 1] : maf_act<ACT>(acc[ct][r0 + 1])) : 0.f;
        const half2_t h = {(half_t)lo, (half_t)hi};
        w[ct] = __builtin_bit_cast(uint32_t, h);
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(op);
    if (nvalid >= CT) {
        if (CT == 8) { *reinterpret_cast<u32x4_t*>(o) = (u32x4_t){w[0], w[1], w[2 % CT], w[3 % CT]}; *reinterpret_cast<u32x4_t*>(o + 4) = (u32x4_t){w[4 % CT], w[5 % CT], w[6 % CT], w[7 % CT]}; }
        else if (CT == 6) {                                               // 24 bytes per lane: 8-byte aligned only
            *reinterpret_cast<u32x2_t*>(o) = (u32x2_t){w[0], w[1]}; *reinterpret_cast<u32x2_t*>(o + 2) = (u32x2_t){w[2 % CT], w[3 % CT]}; *reinterpret_cast<u32x2_t*>(o + 4) = (u32x2_t){w[4 % CT], w[5 % CT]};
        } else if (CT == 4) *reinterpret_cast<u32x4_t*>(o) = (u32x4_t){w[0], w[1], w[2 % CT], w[3 % CT]};
        else *reinterpret_cast<u32x2_t*>(o) = (u32x2_t){w[0], w[1 % CT]};
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct < nvalid) o[ct] = w[ct];
    }
}

// NW = waves per workgroup (4; 8 = a.stream_waves: the instantiations whose KS * CT KiB of weights leave room for one or two workgroups per CU only — twice the
// waves behind the same LDS copy of the weights, i.e. twice the activation tiles in flight per CU and half the tiles per wave; conv_stream_lds_w8.hip)
// ST (training, conv_stream_lds_st.hip; single direct source, four waves): the conv is followed by a training-mode BatchNorm whose first pass would read the
// whole output again for sum x and sum x^2.  Here every lane keeps the two sums of ITS channels over the tiles its wave walks — of the values as stored,
// i.e. after the rounding to fp16: what a pass over the tensor would see — and at the end the lane groups are folded with two cross-lane adds, the four waves
// through the (by then idle) weight area of the LDS, and the workgroup adds 2 x 16 CT values to replica blockIdx % stats_R of the BatchNorm's scratch
// (maf_bn_forward_ex(..., stats_ready = 1) folds the replicas): one atomic per channel, statistic and workgroup.
template <int CT, int KS, bool MULTI, int NW = 4, bool ST = false>
__global__ __launch_bounds__(NW * 64) void conv1x1_stream_lds_kernel(const ConvArgs a) {
    static_assert(!ST || (!MULTI && NW == 4), "the statistics epilogue exists for the single-source four-wave form");
    typedef Frag<half_t> F;
    typedef F::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_raw[];
    frag_t* wl = reinterpret_cast<frag_t*>(wl_raw);                      // [KS][CT][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    // The workgroup keeps ONE channel tile.  Which one: the nN workgroups that walk the SAME pixel tiles (one per channel tile) must sit on the same
    // XCD, or every XCD's L2 fetches the activations once more (PMC, round 3: 576 -> 384 on 20 x 20 fetched 48 MB for a 14.7 MB input — nN = 3 —, the
    // 96-channel-tile forms 1.25-1.36x over the family).  Workgroups go to the XCDs round-robin by blockIdx, so with a workgroup count per channel tile
    // that is a multiple of 8 (the launcher rounds it): XCD = blockIdx & 7, inside it channel tiles fastest.
    const int nwg = gridDim.x / a.nN;
    int n_tile, wg;
    if ((nwg & 7) == 0 && !(MAF_KO & 128)) {
        const int j = blockIdx.x >> 3;
        n_tile = j % a.nN;
        wg = (int)(blockIdx.x & 7) + 8 * (j / a.nN);
    } else {
        n_tile = blockIdx.x % a.nN;
        wg = blockIdx.x / a.nN;
    }
    const int ntiles = (a.M + 15) >> 4;
    {
        // the channel tile's fragments travel global -> LDS by DMA (global_load_lds: 1 KiB per wave-instruction to a wave-uniform base + lane * 16;
        // no registers, no ds_write, ALL of them in flight at once — the first version walked them with a load -> wait -> ds_write loop, 18
        // dependent L2 round trips per workgroup before the first MFMA)
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)(n_tile * CT) * KS) * 1024;   // [ct][ks][64 x 16 B]
        for (int f = wave; f < KS * CT; f += NW) {                      // LDS order: fragment f = ks * CT + ct
            const int ks = f / CT, ct = f - ks * CT;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsrc + ((size_t)ct * KS + ks) * 1024 + lane * 16),
                                             (void __attribute__((address_space(3)))*)(wl_raw + f * 1024), 16, 0, 0);
        }
    }
    const int cl = n_tile * (16 * CT) + p * CT;
    float bias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bias[ct] = a.bias[cl + ct];
    const int nvalid = a.Cout - cl;

    bool up_any = false;
    if constexpr (MULTI) up_any = a.srcMode[0] == MAF_SRC_UP2 || (a.nsrc > 1 && a.srcMode[1] == MAF_SRC_UP2) || (a.nsrc > 2 && a.srcMode[2] == MAF_SRC_UP2) || (a.nsrc > 3 && a.srcMode[3] == MAF_SRC_UP2);
    // per k-step: which source, which channel chunk (scalar; steps past a source's last chunk read chunk 0 x zero weights)
    auto load_tile = [&](int t, frag_t (&af)[KS]) {
        if (MAF_KO & 2) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = (frag_t)(half_t)(0.001f * (float)(lane + ks + t));
            return;
        }
        int m = t * 16 + p;
        m = m < a.M ? m : a.M - 1;
        if constexpr (!MULTI) {
            if (a.srcMode[0] == MAF_SRC_DIRECT) {                        // (uniform)
                const half_t* q = static_cast<const half_t*>(a.src[0]) + (size_t)m * a.srcStride[0] + a.srcCoff[0];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    int c = ks * 32 + g * 8;
                    c = c < a.Cin ? c : 0;
                    af[ks] = ldg16<half_t>(q + c);
                }
            } else {
                // MAF_SRC_POOL2 (MPRep's MaxPool2d(2, 2) in front of its 1x1, common.py:776-792): the source grid is 2H x 2W and a fragment is the
                // element-wise maximum of the four pixels of its window; MAF_SRC_SUB2: the window's top-left pixel alone
                const int x = m % a.W, tq = m / a.W, y = tq % a.H, bb = tq / a.H;
                const half_t* q = static_cast<const half_t*>(a.src[0]) + (size_t)((size_t)(bb * 2 * a.H + 2 * y) * (2 * a.W) + 2 * x) * a.srcStride[0] + a.srcCoff[0];
                const bool one = a.srcMode[0] == MAF_SRC_SUB2;
                const size_t cs = one ? 0 : (size_t)a.srcStride[0], rs = one ? 0 : (size_t)(2 * a.W) * a.srcStride[0];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    int c = ks * 32 + g * 8;
                    c = c < a.Cin ? c : 0;
                    const frag_t v0 = ldg16<half_t>(q + c), v1 = ldg16<half_t>(q + cs + c), v2 = ldg16<half_t>(q + rs + c), v3 = ldg16<half_t>(q + rs + cs + c);
                    af[ks] = F::vmax(F::vmax(v0, v1), F::vmax(v2, v3));
                }
            }
        } else {
            size_t pix_up = (size_t)m;                                   // the pixel of a nearest-neighbour up-sampled source: only worked out (three
            if (up_any) {                                                // divisions per tile) when the launch has one (uniform)
                const int x = m % a.W, tq = m / a.W, y = tq % a.H, bb = tq / a.H;
                pix_up = (size_t)((bb * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1));
            }
            const half_t* q[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half_t* base = static_cast<const half_t*>(a.src[s < a.nsrc ? s : 0]);
                const int si = s < a.nsrc ? s : 0;
                const size_t pix = a.srcMode[si] == MAF_SRC_UP2 ? pix_up : (size_t)m;
                q[s] = base + pix * a.srcStride[si] + a.srcCoff[si];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int i1 = ks >= a.cum[1], i2 = ks >= a.cum[2], i3 = ks >= a.cum[3];
                const int first = i3 ? a.cum[3] : i2 ? a.cum[2] : i1 ? a.cum[1] : 0;
                const int srcC = i3 ? a.srcC[3] : i2 ? a.srcC[2] : i1 ? a.srcC[1] : a.srcC[0];
                const half_t* qq = i3 ? q[3] : i2 ? q[2] : i1 ? q[1] : q[0];
                int c = (ks - first) * 32 + g * 8;
                c = c < srcC ? c : 0;
                af[ks] = ldg16<half_t>(qq + c);
            }
        }
    };
    const int act = a.act;                                               // uniform: the activation is picked once per tile, not per value
    float ssum[ST ? CT : 1], ssq[ST ? CT : 1];
#pragma unroll
    for (int ct = 0; ct < (ST ? CT : 1); ++ct) ssum[ct] = ssq[ct] = 0.f;
    uint32_t wbase[3];                                                   // LDS addresses of this lane's 16 bytes in fragment 0, 64, 128
#pragma unroll
    for (int k = 0; k < 3; ++k) wbase[k] = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(wl_raw + k * 65536 + lane * 16);
    auto compute_store = [&](int t, const frag_t (&af)[KS]) {
        f32x4_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4_t){bias[ct], bias[ct], bias[ct], bias[ct]};
        if constexpr ((MAF_KO & 9) != 0) {                                // knock-out builds: the plain form
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                frag_t wf[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wf[ct] = (MAF_KO & 1) ? (frag_t)(half_t)(0.002f * (float)(lane + ct)) : wl[(ks * CT + ct) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (MAF_KO & 8) acc[ct][0] += (float)af[ks][0] + (float)wf[ct][0];
                    else acc[ct] = F::mma(af[ks], wf[ct], acc[ct]);
                }
            }
        } else {
            // KS * CT steps (k-step, channel tile), ONE straight line, software-pipelined by hand: the fragment of step t + RD is read before the MFMA of
            // step t (LDS returns in order: when step t is consumed the min(RD, steps left) reads issued after its own may still be in flight).  The
            // offset field of a DS instruction holds 64 KiB: one base register per 64 KiB of the (up to 160 KiB) fragment array.
            constexpr int NSTEP = KS * CT, RD = NSTEP > 6 ? 6 : NSTEP - 1;
            u32x4_t wr[RD + 1];
            auto ld_step = [&](auto idx) {
                constexpr int s_ = decltype(idx)::value;
                if constexpr (s_ < NSTEP) sl_ds_read_b128<(s_ * 1024) % 65536>(wr[s_ % (RD + 1)], wbase[(s_ * 1024) / 65536]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the counter now counts only the reads below
            sl_static_for<RD>([&](auto idx) { ld_step(idx); });
            sl_static_for<NSTEP>([&](auto idx) {
                constexpr int s_ = decltype(idx)::value, ks = s_ / CT, ct = s_ % CT, sl = s_ % (RD + 1);
                ld_step(std::integral_constant<int, s_ + RD>{});
                constexpr int ahead = (NSTEP - 1 - s_) < RD ? (NSTEP - 1 - s_) : RD;
                sl_wait_lgkm<ahead>(wr[sl]);
                acc[ct] = F::mma(af[ks], __builtin_bit_cast(frag_t, wr[sl]), acc[ct]);
                __builtin_amdgcn_sched_barrier(0);                      // keep the issue order as written
            });
        }
        if (a.out_pairs) {                                                // pixel pairs for a depth-wise consumer (csrc/dwconv_p2.hip); M is even
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += 2) {
                const int m = t * 16 + g * 4 + r0;
                if (m >= a.M) continue;
                if ((MAF_KO & 4) && acc[0][r0] != 12345.678f) continue;
                half_t* op = static_cast<half_t*>(a.out) + (size_t)m * a.out_stride + (size_t)(a.out_coff + cl) * 2;
                const bool second = m + 1 < a.M;
                if (act == MAF_ACT_SILU) sl_store_pair<MAF_ACT_SILU, CT>(acc, r0, op, nvalid, second);
                else if (act == MAF_ACT_NONE) sl_store_pair<MAF_ACT_NONE, CT>(acc, r0, op, nvalid, second);
                else if (act == MAF_ACT_RELU) sl_store_pair<MAF_ACT_RELU, CT>(acc, r0, op, nvalid, second);
                else sl_store_pair<MAF_ACT_SIGMOID, CT>(acc, r0, op, nvalid, second);
            }
            return;
        }
        if constexpr (ST) {                                               // no activation in front of a BatchNorm (launcher); out-of-range rows add nothing
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = t * 16 + g * 4 + r;
                if (m >= a.M) continue;
                half_t* op = static_cast<half_t*>(a.out) + (size_t)m * a.out_stride + a.out_coff + cl;
                half_t hv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    hv[ct] = (half_t)acc[ct][r];
                    const float f = (float)hv[ct];
                    ssum[ct] += f;
                    ssq[ct] = __builtin_fmaf(f, f, ssq[ct]);
                }
                if (nvalid >= CT) {
                    uint32_t w[CT / 2];
#pragma unroll
                    for (int c2 = 0; c2 < CT / 2; ++c2) w[c2] = __builtin_bit_cast(uint32_t, (half2_t){hv[2 * c2], hv[2 * c2 + 1]});
                    if (CT == 8) *reinterpret_cast<u32x4_t*>(op) = (u32x4_t){w[0], w[1], w[2], w[3 % (CT / 2)]};
                    else if (CT == 6) { *reinterpret_cast<u32x2_t*>(op) = (u32x2_t){w[0], w[1]}; *reinterpret_cast<uint32_t*>(op + 4) = w[2 % (CT / 2)]; }
                    else if (CT == 4) *reinterpret_cast<u32x2_t*>(op) = (u32x2_t){w[0], w[1 % (CT / 2)]};
                    else *reinterpret_cast<uint32_t*>(op) = w[0];
                } else {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (ct < nvalid) op[ct] = hv[ct];
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = t * 16 + g * 4 + r;
            if (m >= a.M) continue;
            if ((MAF_KO & 4) && acc[0][r] != 12345.678f) continue;
            half_t* op = static_cast<half_t*>(a.out) + (size_t)m * a.out_stride + a.out_coff + cl;
            if (act == MAF_ACT_SILU) sl_store<MAF_ACT_SILU, CT>(acc, r, op, nvalid);
            else if (act == MAF_ACT_NONE) sl_store<MAF_ACT_NONE, CT>(acc, r, op, nvalid);
            else if (act == MAF_ACT_RELU) sl_store<MAF_ACT_RELU, CT>(acc, r, op, nvalid);
            else sl_store<MAF_ACT_SIGMOID, CT>(acc, r, op, nvalid);
        }
    };

    frag_t fa[KS];
    const int stride = nwg * NW;
    int t = wg * NW + wave;
    const bool any = t < ntiles;
    if (any) load_tile(t, fa);                                           // in flight beside the weight DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's DMA pieces have landed ...
    __syncthreads();                                                     // ... and everybody else's
    if constexpr (ST) {
        frag_t fb2[KS];
        if (any) {
            while (true) {
                const int t1 = t + stride;
                if (t1 < ntiles) load_tile(t1, fb2);
                compute_store(t, fa);
                if (t1 >= ntiles) break;
                const int t2 = t1 + stride;
                if (t2 < ntiles) load_tile(t2, fa);
                compute_store(t1, fb2);
                if (t2 >= ntiles) break;
                t = t2;
            }
        }
        // lane (g, p) holds the sums of channels cl .. cl + CT - 1 over rows 4g .. 4g + 3 of its tiles: fold the four lane groups, then the four waves
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            ssum[ct] += __shfl_xor(ssum[ct], 16); ssum[ct] += __shfl_xor(ssum[ct], 32);
            ssq[ct] += __shfl_xor(ssq[ct], 16); ssq[ct] += __shfl_xor(ssq[ct], 32);
        }
        __syncthreads();                                                 // nobody reads weight fragments any more
        float* red = reinterpret_cast<float*>(wl_raw);                   // [4 waves][2][16 CT]  (KS * CT KiB >= 2 KiB: fits for every instantiation)
        static_assert(KS * CT * 1024 >= 4 * 2 * 16 * CT * 4, "reduction area inside the weight area");
        if (g == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                red[(wave * 2 + 0) * 16 * CT + p * CT + ct] = ssum[ct];
                red[(wave * 2 + 1) * 16 * CT + p * CT + ct] = ssq[ct];
            }
        }
        __syncthreads();
        if (tid < 2 * 16 * CT) {
            const int which = tid / (16 * CT), j = tid - which * 16 * CT, c = n_tile * 16 * CT + j;
            const float v = red[(0 * 2 + which) * 16 * CT + j] + red[(1 * 2 + which) * 16 * CT + j] + red[(2 * 2 + which) * 16 * CT + j] + red[(3 * 2 + which) * 16 * CT + j];
            if (c < a.Cout) atomicAdd(a.stats + ((size_t)(blockIdx.x % a.stats_R) * 2 + which) * a.Cout + c, v);
        }
        return;
    }
    if (!any) return;
    if constexpr (NW == 8) {
        // two waves per SIMD: the partner's multiplies cover this wave's loads — ONE fragment set per wave (two sets of up to 24 fragments do not fit the
        // 256 registers a wave of a 512-thread workgroup gets)
        while (true) {
            compute_store(t, fa);
            t += stride;
            if (t >= ntiles) break;
            load_tile(t, fa);
        }
        return;
    }
    frag_t fb[KS];
    while (true) {
        const int t1 = t + stride;
        if (t1 < ntiles) load_tile(t1, fb);
        compute_store(t, fa);
        if (t1 >= ntiles) break;
        const int t2 = t1 + stride;
        if (t2 < ntiles) load_tile(t2, fa);
        compute_store(t1, fb);
        if (t2 >= ntiles) break;
        t = t2;
    }
}

template <int CT, int KS, bool MULTI, int NW = 4, bool ST = false>
int launch_sl(const ConvArgs& a, hipStream_t s) {
    constexpr int lds = KS * CT * 1024;
    static_assert(lds <= 160 * 1024, "weights of one channel tile must fit LDS");
    static int occ = 0;                                                  // resident workgroups per CU of this instantiation (LDS and registers)
    if (!occ) {
        if (lds > 64 * 1024) {
            int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_stream_lds_kernel<CT, KS, MULTI, NW, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(conv1x1_stream_lds)");
            if (rc) return rc;
        }
        int nb = 0;
        int rc = maf_check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv1x1_stream_lds_kernel<CT, KS, MULTI, NW, ST>, NW * 64, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor(conv1x1_stream_lds)");
        if (rc) return rc;
        occ = nb < 1 ? 1 : nb > 4 ? 4 : nb;
    }
    // ONE round of persistent workgroups: exactly what is resident at once (a second round pays the weight DMA and the ramp-up again for a
    // handful of tiles per wave), spread evenly over the channel tiles
    const int ntiles = (a.M + 15) >> 4;
    int per = (ntiles + NW - 1) / NW;                                    // workgroups per channel tile that still have work
    const int cap = occ * 256 / a.nN > 0 ? occ * 256 / a.nN : 1;
    if (per > cap) per = cap;
    if (per >= 8 && a.nN > 1) per &= ~7;                                 // whole XCD rounds: the channel tiles of a pixel tile then share an XCD's L2 (see the kernel)
    hipLaunchKernelGGL((conv1x1_stream_lds_kernel<CT, KS, MULTI, NW, ST>), dim3(per * a.nN), dim3(NW * 64), lds, s, a);
    return maf_check_hip(hipGetLastError(), "conv1x1_stream_lds launch");
}

}  // namespace
