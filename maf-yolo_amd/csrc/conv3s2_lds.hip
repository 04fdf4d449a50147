// 3x3 stride-2 pad-1 conv for narrow layers on big maps (tile_k = 6 of MAF_OP_CONV3X3S2): RepVGGBlock / ConvWrapper / MPRep.conv2 in deploy
// form (yolov6/layers/common.py:216-217, 76-83, 776-792) where 9 * Cin * Cout weights fit the LDS — backbone.3.conv2 (48 -> 48) and
// backbone.18 (48 -> 64) of MAF-YOLO-n on the 160 x 160 map.  The generic template reads every input pixel 2.25 times through L2 and its
// weight fragments once per wave; here a persistent workgroup keeps ALL weight fragments in LDS and stages the (2 TY + 1) x 33 input
// patch of a TY x 16 output tile once (aligned 16-byte loads of the NHWC pixels, prefetched a tile ahead into registers, stored
// pixel-major with a 2 * odd dword stride so the stride-2 fragment reads of 16 lanes cover all 64 banks).  The product is taken as in
// csrc/stem2.hip phase C: K = 9 taps x Cin / 8 (tap, 8-channel group) pairs, four pairs per MFMA k-step, transposed (A = weights,
// B = patch) so a lane holds 4 consecutive output channels of one pixel; bias + activation, through LDS, whole NHWC pixels out.
// C1 > 0: MPRep in ONE launch (common.py:776-792: cat(conv1(MaxPool2d(2, 2)(x)), conv2(x))): the 2 x 2 windows of the tile's output pixels lie
// inside the staged patch, so the pooled 1x1 + SiLU branch is taken from LDS — its (few) weight fragments live in registers — and its C1 channels
// go to the other half of the output pixels; the input is read from HBM once instead of twice.
#include "maf_common.h"
#include "lds_pipe.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko KO_SRCS=conv3s2_lds.hip): 128 = plain round-robin tile order
#endif

namespace {

struct C3Args {
    const half_t* in; const char* rec; half_t* out;
    int B, Hin, Win, H, W, in_stride, in_coff, out_stride, out_coff, act, tilesX, tilesY, ntiles;
    int out1_coff;                                               // C1 > 0: channel offset of the pooled branch's output
};

template <int CIN, int COUT, int TY, int C1>
__global__ __launch_bounds__(256, (9 * CIN * COUT * 2 + (2 * TY + 1) * 33 * ((CIN / 2 + 2) | 2) * 4 + TY * 16 * COUT * 2) <= 80 * 1024 ? 2 : 1) void conv3s2_lds_kernel(const C3Args a) {
    static_assert(C1 % 16 == 0 && C1 <= COUT, "the pooled branch is staged through the conv's own output stage");
    constexpr int KS1 = (CIN + 31) / 32, NT1 = C1 / 16;
    constexpr int MR = TY / 4, TX = 16, SR = 2 * TY + 1, SC = 2 * TX + 1, SP = SR * SC;
    constexpr int GR = CIN / 8, NP = 9 * GR, KS = (NP + 3) / 4, NT = COUT / 16;
    constexpr int TS = (CIN / 2 + 2) | 2;                        // pixel stride in dwords: 2 * odd
    static_assert((TS / 2) % 2 == 1 && TS * 2 >= CIN, "patch stride");
    constexpr int TSH = TS * 2, WB = KS * NT * 64 * 16;
    constexpr int NCHK = SP * GR, PF = (NCHK + 255) / 256;
    __shared__ __attribute__((aligned(16))) half_t s_T[SP * TSH + 64];
    __shared__ __attribute__((aligned(16))) half_t s_out[TY * TX * COUT];
    __shared__ __attribute__((aligned(16))) char s_w[WB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
    // weight fragments: global -> LDS by DMA (1 KiB per wave-instruction, every piece in flight at once; published by the first barrier of the tile loop)
    for (int v0 = wave * 64; v0 < WB / 16; v0 += 256)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.rec + (size_t)(v0 + lane) * 16),
                                         (void __attribute__((address_space(3)))*)(s_w + v0 * 16), 16, 0, 0);
    const float* bias = reinterpret_cast<const float*>(a.rec + WB);
    f32x4_t bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4_t*>(bias + 16 * t + 4 * g);
    // pooled branch: weight fragments [KS1][NT1][64][8] f16 and the bias behind the conv's record; lane (g, n) of k-step s: output channel 16t + n,
    // input channels 32s + 8g .. + 7 (zero rows past CIN)
    half8_t w1f[KS1 * NT1 > 0 ? KS1 * NT1 : 1];
    f32x4_t b1v[NT1 > 0 ? NT1 : 1];
    if constexpr (C1 > 0) {
        const half8_t* w1g = reinterpret_cast<const half8_t*>(a.rec + WB + COUT * 4);
#pragma unroll
        for (int i = 0; i < KS1 * NT1; ++i) w1f[i] = w1g[i * 64 + lane];
        const float* bias1 = reinterpret_cast<const float*>(a.rec + WB + COUT * 4 + KS1 * NT1 * 1024);
#pragma unroll
        for (int t = 0; t < NT1; ++t) b1v[t] = *reinterpret_cast<const f32x4_t*>(bias1 + 16 * t + 4 * g);
    }
    int off1[KS];                                                // patch offsets (halves) of this lane's (tap, group) pair in every k-step
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int q = min(4 * s + g, NP - 1), tap = q / GR, grp = q - tap * GR;
        off1[s] = ((tap / 3) * SC + tap % 3) * TSH + 8 * grp;
    }
    const half8_t* wf = reinterpret_cast<const half8_t*>(s_w);
    uint4 pf[PF];
    bool pf_in[PF];
    auto prefetch = [&](int tile) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const half_t* img = a.in + (size_t)b * a.Hin * a.Win * a.in_stride + a.in_coff;
        const int iy0 = 2 * ty * TY - 1, ix0 = 2 * tx * TX - 1;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = min(tid + 256 * u, NCHK - 1);
            const int px = e / GR, grp = e - px * GR, sr = px / SC, sc = px - sr * SC;
            const int iy = iy0 + sr, ix = ix0 + sc;
            pf_in[u] = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const size_t at = pf_in[u] ? ((size_t)iy * a.Win + ix) * a.in_stride + 8 * grp : 0;   // clamped: the load stays unconditional
            pf[u] = *reinterpret_cast<const uint4*>(img + at);
        }
    };
    // XCD-contiguous tile order (as csrc/stem2.hip): workgroups go to the 8 XCDs round-robin and neighbouring tiles share halo rows / columns — an XCD
    // walks one contiguous eighth of the tiles, so the shared lines are hits in ITS L2
    int t_first = blockIdx.x, t_end = a.ntiles, t_step = gridDim.x;
    if ((gridDim.x & 7) == 0 && !(MAF_KO & 128)) {
        const int xcd = blockIdx.x & 7, q = a.ntiles >> 3, r = a.ntiles & 7, base = xcd * q + min(xcd, r);
        t_first = base + (int)(blockIdx.x >> 3); t_end = base + q + (xcd < r ? 1 : 0); t_step = gridDim.x >> 3;
    }
    if (t_first < t_end) prefetch(t_first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the DMA pieces of this wave have landed (the barrier below publishes everybody's)
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX, ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const int Y0 = ty * TY, X0 = tx * TX;
        __syncthreads();                                         // s_T and s_out of the previous tile are free; first pass: s_w is in place
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = tid + 256 * u;
            if (e < NCHK) {
                const int px = e / GR, grp = e - px * GR;
                uint4 v = pf[u];
                if (!pf_in[u]) v = make_uint4(0u, 0u, 0u, 0u);   // zero padding
                uint2* d = reinterpret_cast<uint2*>(s_T + px * TSH + 8 * grp);   // 8-byte aligned (the pixel stride is 8 * odd bytes)
                d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
            }
        }
        __syncthreads();
        if (tile + t_step < t_end) prefetch(tile + t_step);
        f32x4_t acc[MR][NT];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            // KS k-steps, one straight line, the reads of step s + RD (two 8-byte halves of the patch fragment per m-tile, NT weight fragments) issued
            // before the MFMAs of step s (lds_pipe.h: as plain loads every step waited a full LDS round trip in front of its MFMAs)
            constexpr int RD = 2, PER = 2 * MR + NT;                          // reads per step
            u32x2_t plo[RD + 1][MR], phi[RD + 1][MR];
            u32x4_t wr[RD + 1][NT];
            const uint32_t wa = lp_lds_addr(s_w + lane * 16);
            uint32_t pa[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) pa[m] = lp_lds_addr(s_T + ((2 * (wave * MR + m)) * SC + 2 * n) * TSH);
            auto ld_step = [&](auto idx) {
                constexpr int s_ = decltype(idx)::value;
                if constexpr (s_ < KS) {
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const uint32_t ad = pa[m] + (uint32_t)off1[s_] * 2;
                        asm volatile("ds_read_b64 %0, %1" : "=v"(plo[s_ % (RD + 1)][m]) : "v"(ad));
                        asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(phi[s_ % (RD + 1)][m]) : "v"(ad));
                    }
                    lp_static_for<NT>([&](auto tt) {
                        constexpr int t = decltype(tt)::value;
                        lp_ds_read_b128<((s_ * NT + t) * 1024) % 65536>(wr[s_ % (RD + 1)][t], wa + ((s_ * NT + t) * 1024) / 65536 * 65536);
                    });
                }
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lp_static_for<RD>([&](auto idx) { ld_step(idx); });
            lp_static_for<KS>([&](auto idx) {
                constexpr int s_ = decltype(idx)::value, sl = s_ % (RD + 1);
                ld_step(std::integral_constant<int, s_ + RD>{});
                constexpr int ahead = ((KS - 1 - s_) < RD ? (KS - 1 - s_) : RD) * PER;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(ahead < 15 ? ahead : 15) : "memory");
                half8_t tf[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    asm volatile("" : "+v"(plo[sl][m]), "+v"(phi[sl][m]));    // consumed after the wait
                    const u32x4_t v4 = {plo[sl][m][0], plo[sl][m][1], phi[sl][m][0], phi[sl][m][1]};
                    tf[m] = __builtin_bit_cast(half8_t, v4);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    asm volatile("" : "+v"(wr[sl][t]));
                    const half8_t w8 = __builtin_bit_cast(half8_t, wr[sl][t]);
#pragma unroll
                    for (int m = 0; m < MR; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w8, tf[m], acc[m][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // the activation is picked once per tile (a switch per VALUE made the epilogue a chain of branches)
        auto stage_out = [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    half4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (half_t)maf_act<ACT>(acc[m][t][q] + bv[t][q]);
                    *reinterpret_cast<half4_t*>(s_out + ((wave * MR + m) * TX + n) * COUT + 16 * t + 4 * g) = v;
                }
        };
        if (a.act == MAF_ACT_SILU) stage_out(std::integral_constant<int, MAF_ACT_SILU>{});
        else if (a.act == MAF_ACT_RELU) stage_out(std::integral_constant<int, MAF_ACT_RELU>{});
        else if (a.act == MAF_ACT_NONE) stage_out(std::integral_constant<int, MAF_ACT_NONE>{});
        else stage_out(std::integral_constant<int, MAF_ACT_SIGMOID>{});
        // a wave copies out the MR rows it staged itself: its LDS operations execute in order, no workgroup barrier (the one at the top of the
        // tile loop still separates this tile's readers of s_T from the next tile's writers)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int CPP = COUT / 8;
        for (int ql = lane; ql < MR * TX * CPP; ql += 64) {
            const int px = wave * MR * TX + ql / CPP, part = ql % CPP;
            const int oy = Y0 + px / TX, ox = X0 + px % TX;
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<uint4*>(a.out + ((size_t)(b * a.H + oy) * a.W + ox) * a.out_stride + a.out_coff + 8 * part) =
                    *reinterpret_cast<const uint4*>(s_out + px * COUT + 8 * part);
        }
        if constexpr (C1 > 0) {
            // ---- the pooled branch: SiLU(W1 . max over the 2 x 2 window + b1).  Output pixel (r, n) of the tile <- patch pixels (2r + 1 + dy, 2n + 1 + dx)
            // (the patch starts one pixel up and left of the tile's first window); the patch is still in place: the next tile's writers wait behind
            // the barrier at the top of the loop.
            f32x4_t acc1[MR][NT1];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
#pragma unroll
                for (int t = 0; t < NT1; ++t) acc1[m][t] = b1v[t];
                const half_t* p00 = s_T + ((2 * (wave * MR + m) + 1) * SC + 2 * n + 1) * TSH;
#pragma unroll
                for (int s_ = 0; s_ < KS1; ++s_) {
                    const int gg = min(4 * s_ + g, GR - 1);          // groups past CIN meet zero weights; stay inside the pixel's CIN halves
                    half8_t v[4];
#pragma unroll
                    for (int w_ = 0; w_ < 4; ++w_) {
                        const uint2* q = reinterpret_cast<const uint2*>(p00 + ((w_ >> 1) * SC + (w_ & 1)) * TSH + 8 * gg);   // 8-byte aligned
                        const uint2 lo = q[0], hi = q[1];
                        v[w_] = __builtin_bit_cast(half8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                    const half8_t mx = __builtin_elementwise_max(__builtin_elementwise_max(v[0], v[1]), __builtin_elementwise_max(v[2], v[3]));   // v_pk_max_f16
#pragma unroll
                    for (int t = 0; t < NT1; ++t) acc1[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[s_ * NT1 + t], mx, acc1[m][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's copy-out reads of the conv's rows are done before it restages
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    half4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (half_t)maf_act<MAF_ACT_SILU>(acc1[m][t][q]);
                    *reinterpret_cast<half4_t*>(s_out + ((wave * MR + m) * TX + n) * COUT + 16 * t + 4 * g) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int CPP1 = C1 / 8;
            for (int ql = lane; ql < MR * TX * CPP1; ql += 64) {
                const int px = wave * MR * TX + ql / CPP1, part = ql % CPP1;
                const int oy = Y0 + px / TX, ox = X0 + px % TX;
                if (oy < a.H && ox < a.W)
                    *reinterpret_cast<uint4*>(a.out + ((size_t)(b * a.H + oy) * a.W + ox) * a.out_stride + a.out1_coff + 8 * part) =
                        *reinterpret_cast<const uint4*>(s_out + px * COUT + 8 * part);
            }
        }
    }
}

}  // namespace

extern "C" int64_t maf_conv3s2_lds_record_bytes(int32_t Cin, int32_t Cout) {
    const int ks = (9 * (Cin / 8) + 3) / 4;
    return (int64_t)ks * (Cout / 16) * 64 * 16 + Cout * 4;
}

// ... with the pooled 1x1 branch of MPRep (C1 output channels) behind it: fragments [ceil(Cin / 32)][C1 / 16][64][8] f16 | bias fp32 [C1]
extern "C" int64_t maf_mprep_lds_record_bytes(int32_t Cin, int32_t Cout, int32_t C1) {
    return maf_conv3s2_lds_record_bytes(Cin, Cout) + (int64_t)((Cin + 31) / 32) * (C1 / 16) * 64 * 16 + C1 * 4;
}

int maf_launch_conv3s2_lds(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16 && !op->out_f32, "conv3x3s2 (tile_k = 6): fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && sr.C == op->Cin, "conv3x3s2 (tile_k = 6): one direct source");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "conv3x3s2 (tile_k = 6): 16-byte aligned channel slices");
    MAF_REQUIRE(op->w && op->out, "conv3x3s2 (tile_k = 6): null pointer");
    MAF_REQUIRE(op->Hin > 0 && op->Win > 0 && (op->Hin - 1) / 2 + 1 == op->H && (op->Win - 1) / 2 + 1 == op->W, "conv3x3s2: H,W must equal floor((Hin-1)/2)+1");
    MAF_REQUIRE(op->act >= 0 && op->act <= 3, "conv3x3s2: bad act");
    C3Args a;
    a.in = static_cast<const half_t*>(sr.ptr); a.rec = static_cast<const char*>(op->w); a.out = static_cast<half_t*>(op->out);
    a.B = op->B; a.Hin = op->Hin; a.Win = op->Win; a.H = op->H; a.W = op->W;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff; a.act = op->act;
    a.out1_coff = op->reg_stride;
    if (op->nc) {                                                 // MPRep: the MaxPool2d(2, 2) + 1x1 + SiLU branch rides along (nc = its channels, reg_stride = where they go)
        MAF_REQUIRE((op->Cin == 48 || op->Cin == 64) && op->Cout == op->Cin && op->nc == op->Cin, "conv3x3s2 (tile_k = 6): the pooled 1x1 branch exists for 48 -> 48 + 48 and 64 -> 64 + 64");
        MAF_REQUIRE(op->Hin == 2 * op->H && op->Win == 2 * op->W, "conv3x3s2 (tile_k = 6) with the pooled branch: even input size (MaxPool2d(2, 2) windows)");
        MAF_REQUIRE(op->reg_stride >= 0 && op->reg_stride % 8 == 0 && (op->reg_stride + op->nc <= op->out_coff || op->reg_stride >= op->out_coff + op->Cout)
                    && op->reg_stride + op->nc <= op->out_stride, "conv3x3s2 (tile_k = 6): the pooled branch's channels (reg_stride = their offset) must lie beside the conv's");
    }
    const int ty = 4;                                             // 4 x 16 output tiles
    a.tilesX = maf_cdiv(a.W, 16); a.tilesY = maf_cdiv(a.H, ty); a.ntiles = a.B * a.tilesX * a.tilesY;
    const dim3 grid(std::min(a.ntiles, op->tile_c > 0 ? op->tile_c * 64 : 256)), blk(256);
#define MAF_C3(CI, CO) hipLaunchKernelGGL((conv3s2_lds_kernel<CI, CO, 4, 0>), grid, blk, 0, s, a)
    if (op->nc == 48) hipLaunchKernelGGL((conv3s2_lds_kernel<48, 48, 4, 48>), grid, blk, 0, s, a);
    else if (op->nc == 64) hipLaunchKernelGGL((conv3s2_lds_kernel<64, 64, 4, 64>), grid, blk, 0, s, a);
    else if (op->Cin == 48 && op->Cout == 48) MAF_C3(48, 48);
    else if (op->Cin == 48 && op->Cout == 64) MAF_C3(48, 64);
    else if (op->Cin == 64 && op->Cout == 64) MAF_C3(64, 64);
    else { maf_set_error("conv3x3s2 (tile_k = 6): (Cin, Cout) must be (48, 48), (48, 64) or (64, 64)"); return MAF_E_UNSUPPORTED; }
#undef MAF_C3
    return maf_check_hip(hipGetLastError(), "conv3s2_lds launch");
}
