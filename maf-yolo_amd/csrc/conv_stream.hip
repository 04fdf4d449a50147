// Streaming 1x1 convolution (single direct source, short reduction: Cin <= 128) with *persistent* waves.
//
// Same reference code as conv_mfma.inc.h (Conv.forward_fuse, yolov6/layers/common.py:49-50) and the same operand layout
// (activations = MFMA A operand: one 16-byte global load per lane; weights = B operand, host-packed fragments; lane (g, p) owns
// CT consecutive output channels of 4 pixels).  What differs is the schedule.  In the one-tile-per-wave kernel a wave loads,
// waits the full memory latency, computes a few MFMAs, stores and dies: between its stores and the first load of the wave
// that replaces it nothing is in flight for that slot.  Here a wave walks many pixel tiles: its weight fragments and bias
// stay in registers, and the activation fragments of tile t+1 are already in flight while tile t is multiplied, activated
// and stored — the loads of one tile overlap the epilogue of the previous one by construction, not by occupancy.
#include "conv_mfma.inc.h"

namespace {

template <int PT, int CT, int KS>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const ConvArgs a) {
    typedef Frag<half_t> F;
    typedef F::type frag_t;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;          // persistent waves
    const int n_tile = gw % a.nN;                                       // a wave keeps ONE channel tile: weights loaded once
    const int w0 = gw / a.nN, wstride = nw / a.nN;                      // host launches a multiple of nN waves
    const int ntiles = (a.M + 16 * PT - 1) / (16 * PT);
    const half_t* s0 = static_cast<const half_t*>(a.src[0]) + a.srcCoff[0];

    frag_t wf[KS][CT];
    const frag_t* wbase = reinterpret_cast<const frag_t*>(a.w) + ((size_t)(n_tile * CT) * KS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wf[ks][ct] = wbase[((size_t)ct * KS + ks) * 64];
    const int cl = n_tile * (16 * CT) + p * CT;
    float bias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bias[ct] = a.bias[cl + ct];
    const int nvalid = a.Cout - cl;
    int coff[KS];                                                      // channel chunk of this lane per k-step (past the end: chunk 0 x zero weights)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { const int c = ks * 32 + g * 8; coff[ks] = c < a.Cin ? c : 0; }

    auto load_tile = [&](int t, frag_t (&af)[PT][KS]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int m = t * (16 * PT) + pt * 16 + p;
            m = m < a.M ? m : a.M - 1;                                  // rows past the end are never stored
            const half_t* q = s0 + (size_t)m * a.srcStride[0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[pt][ks] = ldg16<half_t>(q + coff[ks]);
        }
    };
    auto compute_store = [&](int t, const frag_t (&af)[PT][KS]) {
        f32x4_t acc[PT][CT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[pt][ct] = (f32x4_t){bias[ct], bias[ct], bias[ct], bias[ct]};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[pt][ct] = F::mma(af[pt][ks], wf[ks][ct], acc[pt][ct]);
            }
        // the activation is picked once per tile, not per value (a per-value switch turns the epilogue into a chain of branches)
        auto epilogue = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = t * (16 * PT) + pt * 16 + g * 4 + r;
                if (m >= a.M) continue;
                float v[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) v[ct] = maf_act<ACT>(acc[pt][ct][r]);
                half_t* op = static_cast<half_t*>(a.out) + (size_t)m * a.out_stride + a.out_coff + cl;
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
        }
        };
        if (a.act == MAF_ACT_SILU) epilogue(std::integral_constant<int, MAF_ACT_SILU>{});
        else if (a.act == MAF_ACT_NONE) epilogue(std::integral_constant<int, MAF_ACT_NONE>{});
        else if (a.act == MAF_ACT_RELU) epilogue(std::integral_constant<int, MAF_ACT_RELU>{});
        else epilogue(std::integral_constant<int, MAF_ACT_SIGMOID>{});
    };

    // two register sets alternate: tile t is consumed from one while tile t + wstride lands in the other
    frag_t fa[PT][KS], fb[PT][KS];
    int t = w0;
    if (t >= ntiles) return;
    load_tile(t, fa);
    while (true) {
        const int t1 = t + wstride;
        if (t1 < ntiles) load_tile(t1, fb);
        compute_store(t, fa);
        if (t1 >= ntiles) break;
        const int t2 = t1 + wstride;
        if (t2 < ntiles) load_tile(t2, fa);
        compute_store(t1, fb);
        if (t2 >= ntiles) break;
        t = t2;
    }
}

template <int PT, int CT, int KS>
int launch_stream(const ConvArgs& a, hipStream_t s) {
    // persistent grid: enough waves to fill the chip several times over (each CU holds <= 8 workgroups of this kernel), a
    // multiple of nN so that every wave keeps one channel tile
    const int ntiles = (a.M + 16 * PT - 1) / (16 * PT);
    int wgs = 256 * 6;
    const int need = (ntiles * a.nN + 3) / 4;
    if (wgs > need) wgs = need;
    int waves = wgs * 4;
    waves = (waves + a.nN - 1) / a.nN * a.nN;
    wgs = (waves + 3) / 4;
    while ((wgs * 4) % a.nN) ++wgs;
    hipLaunchKernelGGL((conv1x1_stream_kernel<PT, CT, KS>), dim3(wgs), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "conv1x1_stream launch");
}

template <int PT, int CT>
int launch_stream_ks(const ConvArgs& a, hipStream_t s) {
    switch (a.ksteps) {
        case 1: return launch_stream<PT, CT, 1>(a, s);
        case 2: return launch_stream<PT, CT, 2>(a, s);
        case 3: if constexpr (CT <= 4) return launch_stream<PT, CT, 3>(a, s); break;
        case 4: if constexpr (CT <= 4) return launch_stream<PT, CT, 4>(a, s); break;
    }
    maf_set_error("conv: tile_k = 3 (persistent streaming 1x1) needs ksteps * tile_c <= 16");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

int maf_conv1x1_stream(const ConvArgs& a, int pt, int ct, hipStream_t s) {
#define MAF_ST(P, C) if (pt == P && ct == C) return launch_stream_ks<P, C>(a, s);
    MAF_ST(1, 2) MAF_ST(2, 2) MAF_ST(1, 4) MAF_ST(2, 4) MAF_ST(1, 6) MAF_ST(2, 6) MAF_ST(1, 8) MAF_ST(2, 8)
#undef MAF_ST
    maf_set_error("conv: tile_k = 3 supports tile_p in {1,2}, tile_c in {2,4,6,8}");
    return MAF_E_UNSUPPORTED;
}
