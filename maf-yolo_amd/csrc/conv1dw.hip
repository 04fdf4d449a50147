// First half of a DepthBottleneckUni in one launch for ANY width c:  T2 = SiLU(DW_k(SiLU(X*W1 + b1)) + bdw)   (conv1 -> conv2 ->
// act of yolov6/layers/common.py:905-909, deploy form).  The fully fused kernel of bottleneck.hip keeps the accumulators of the
// second 1x1 in registers (c VGPRs per wave), which stops at c = 64; here the 3c-wide T1 still never leaves the CU (6c of the
// bottleneck's 14c bytes per pixel disappear, and one launch), T2 goes to memory and the plain 1x1 kernel finishes the block.
//
// One workgroup = one 16 x 16 output tile x ONE block of 32 mid channels (grid = tiles x blocks: plenty of workgroups even on the
// 20 x 20 maps): phase A = the first 1x1 on the (16+k-1)^2 halo tile by MFMA into LDS planes, phase B = the depth-wise conv as
// block-diagonal Toeplitz MFMAs — both exactly as in bottleneck.hip (operand algebra and bank layout are documented there) — then
// bias + SiLU and 16-byte NHWC stores (lane (g, n) ends with the 8 consecutive channels 8g..8g+7 of 4 pixels: the host permutes
// the block's channels so that set s of lane group g is channel 8g + s).
#include "maf_common.h"

namespace {

struct C1dArgs {
    const half_t* x; half_t* out;
    const unsigned char* par;   // [nMB] records: W1 fragments [2][S1][64] half8 | b1 [32] f32 | Toeplitz [8][K][PARTS][16] half8 | bdw [32] f32
    int B, H, W, Cin, Cmid, S1, nMB, x_stride, x_coff, out_stride, out_coff;
    int tilesX, tilesY, nwg, rec;
};

typedef half_t half4v_t __attribute__((ext_vector_type(4)));

template <int K>
struct C1dCfg {
    static constexpr int P = K / 2, PARTS = K > 5 ? 2 : 1;
    static constexpr int RH = 16 + K - 1, RWC = (16 + K - 1 + 3) & ~3, NHP = RH * RWC, NPT = (NHP + 15) / 16;
    static constexpr int RWP = 24;
    static constexpr int PSB = ((RH * RWP * 2 - 8 + 255) / 256) * 256 + 8;
    static constexpr int PS = PSB / 2;
    static constexpr int NTOE = 8 * K * PARTS * 16;
};

template <int K>
__global__ __launch_bounds__(256) void conv1dw_kernel(const C1dArgs a) {
    typedef C1dCfg<K> Cf;
    constexpr int P = Cf::P, PARTS = Cf::PARTS, RWC = Cf::RWC, NHP = Cf::NHP, NPT = Cf::NPT, RWP = Cf::RWP, PS = Cf::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* T1 = reinterpret_cast<half_t*>(smem_raw);                        // [32 slots][PS]
    unsigned char* rec = smem_raw + 32 * Cf::PSB;
    const int OFF_B1 = 2 * a.S1 * 1024, OFF_TOE = OFF_B1 + 128, OFF_BD = OFF_TOE + Cf::NTOE * 16;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mb = lid % a.nMB;                              // the mid blocks of a tile run back to back: its X halo stays in L2
    int tt = lid / a.nMB;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty * 16, x0 = tx * 16;
    const half_t* xin = a.x + (size_t)b * a.H * a.W * a.x_stride + a.x_coff;

    {   // the block's record -> LDS by DMA (1 KiB per wave-instruction)
        const unsigned char* src = a.par + (size_t)mb * a.rec;
        const int nvec = a.rec >> 4;
        for (int v0 = wave * 64; v0 < nvec; v0 += 256)
            if (v0 + lane < nvec)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)(v0 + lane) * 16),
                                                 (void __attribute__((address_space(3)))*)(rec + v0 * 16), 16, 0, 0);
    }
    __syncthreads();

    // ---- A. T1 = SiLU(X * W1[:, block] + b1) on the halo tile; out-of-image pixels are exact zeros (the depth-wise padding)
    {
        const bool interior = y0 - P >= 0 && x0 - P >= 0 && y0 - P + Cf::RH <= a.H && x0 - P + RWC <= a.W;
        const half8_t* w1l = reinterpret_cast<const half8_t*>(rec) + lane;
        const float b1v0 = reinterpret_cast<const float*>(rec + OFF_B1)[p], b1v1 = reinterpret_cast<const float*>(rec + OFF_B1)[16 + p];
        for (int t = wave; t < NPT; t += 4) {
            const int m = t * 16 + p;
            const int hrp = m / RWC, hcp = m - hrp * RWC;
            const int iy = min(max(y0 - P + hrp, 0), a.H - 1), ix = min(max(x0 - P + hcp, 0), a.W - 1);   // clamped: masked below
            const half_t* px = xin + (size_t)(iy * a.W + ix) * a.x_stride;
            f32x4_t acc1[2] = {(f32x4_t)0.f, (f32x4_t)0.f};
            for (int ks0 = 0; ks0 < a.S1; ks0 += 4) {
                half8_t af[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int c = (ks0 + u) * 32 + g * 8;
                    c = c < a.Cin ? c : 0;                               // chunk past the end: zero weight rows
                    af[u] = *reinterpret_cast<const half8_t*>(px + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ks0 + u < a.S1) {
                        acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], w1l[(0 * a.S1 + ks0 + u) * 64], acc1[0], 0, 0, 0);
                        acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], w1l[(1 * a.S1 + ks0 + u) * 64], acc1[1], 0, 0, 0);
                    }
            }
            const int m0 = t * 16 + g * 4;
            const int hr = m0 / RWC, hc0 = m0 - hr * RWC;
            uint32_t mlo = 0xffffffffu, mhi = 0xffffffffu;
            if (!interior) {
                const int iy2 = y0 - P + hr, ixb = x0 - P + hc0;
                const bool rowok = (unsigned)iy2 < (unsigned)a.H;
                const uint32_t k0 = (rowok && (unsigned)(ixb + 0) < (unsigned)a.W) ? 0x0000ffffu : 0u, k1 = (rowok && (unsigned)(ixb + 1) < (unsigned)a.W) ? 0xffff0000u : 0u;
                const uint32_t k2 = (rowok && (unsigned)(ixb + 2) < (unsigned)a.W) ? 0x0000ffffu : 0u, k3 = (rowok && (unsigned)(ixb + 3) < (unsigned)a.W) ? 0xffff0000u : 0u;
                mlo = k0 | k1; mhi = k2 | k3;
            }
            half_t* dst = T1 + (size_t)p * PS + hr * RWP + hc0;          // column p of channel tile ct -> plane 16*ct + p
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float bv = ct ? b1v1 : b1v0;
                const half2_t h01 = {(half_t)maf_act<MAF_ACT_SILU>(acc1[ct][0] + bv), (half_t)maf_act<MAF_ACT_SILU>(acc1[ct][1] + bv)};
                const half2_t h23 = {(half_t)maf_act<MAF_ACT_SILU>(acc1[ct][2] + bv), (half_t)maf_act<MAF_ACT_SILU>(acc1[ct][3] + bv)};
                const u32x2_t w = {__builtin_bit_cast(uint32_t, h01) & mlo, __builtin_bit_cast(uint32_t, h23) & mhi};
                if (m0 < NHP) *reinterpret_cast<u32x2_t*>(dst + (size_t)(16 * ct) * PS) = w;
            }
        }
    }
    __syncthreads();

    // ---- B. depth-wise conv: lane (g, n = p) reads plane 4s + g
    const bool toe_active = (p >> 2) == g;
    const int q4 = wave * 4;
    int hi_off = 4;
    asm volatile("" : "+v"(hi_off));
    f32x4_t dacc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) dacc[s] = (f32x4_t)0.f;
    const half8_t* tl = reinterpret_cast<const half8_t*>(rec + OFF_TOE) + p;
    const half_t* t1l = T1 + (size_t)g * PS + p * RWP + q4;
    const half_t* t1h = t1l + hi_off;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                half8_t av = (half8_t)(half_t)0;
                if (toe_active) av = tl[((s * K + ky) * PARTS + part) * 16];
                const int o = s * 4 * PS + ky * RWP + part * 4;
                const half4v_t lo = *reinterpret_cast<const half4v_t*>(t1l + o), hi = *reinterpret_cast<const half4v_t*>(t1h + o);
                const half8_t bv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                dacc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, dacc[s], 0, 0, 0);
            }
        }
    }
    // ---- T2: lane (g, n): channels mb*32 + 8g .. +7 of pixels (row y0 + n, x = x0 + 4q + r)
    const int c0 = mb * 32 + g * 8;
    if (c0 >= a.Cmid) return;
    const f32x4_t bd0 = reinterpret_cast<const f32x4_t*>(rec + OFF_BD)[g * 2], bd1 = reinterpret_cast<const f32x4_t*>(rec + OFF_BD)[g * 2 + 1];
    const float bs[8] = {bd0[0], bd0[1], bd0[2], bd0[3], bd1[0], bd1[1], bd1[2], bd1[3]};
    const int oy = y0 + p;
    if (oy >= a.H) return;
    half_t* orow = a.out + ((size_t)b * a.H + oy) * a.W * a.out_stride + a.out_coff + c0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ox = x0 + q4 + r;
        if (ox >= a.W) continue;
        half8_t o;
#pragma unroll
        for (int s = 0; s < 8; ++s) o[s] = (half_t)maf_act<MAF_ACT_SILU>(dacc[s][r] + bs[s]);
        *reinterpret_cast<half8_t*>(orow + (size_t)ox * a.out_stride) = o;
    }
}

template <int K>
int launch_k(const C1dArgs& a, hipStream_t s) {
    const size_t lds = (size_t)32 * C1dCfg<K>::PSB + (size_t)a.rec;
    MAF_REQUIRE(lds <= 160 * 1024, "conv1dw: tile + record do not fit LDS");
    static bool attr = false;
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1dw_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(conv1dw)");
        if (rc) return rc;
        attr = true;
    }
    hipLaunchKernelGGL((conv1dw_kernel<K>), dim3(a.nwg), dim3(256), lds, s, a);
    return maf_check_hip(hipGetLastError(), "conv1dw launch");
}

}  // namespace

extern "C" int64_t maf_conv1dw_record_bytes(int32_t k, int32_t Cin) {
    const int parts = k > 5 ? 2 : 1, s1 = (Cin + 31) / 32;
    return (int64_t)2 * s1 * 1024 + 128 + (int64_t)8 * k * parts * 16 * 16 + 128;
}

int maf_launch_conv1dw(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "conv1dw: fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && op->out && op->w, "conv1dw: one direct source, w = block records");
    MAF_REQUIRE(op->Cin % 8 == 0 && op->Cout % 8 == 0 && op->Cin <= 512, "conv1dw: Cin, Cmid multiples of 8, Cin <= 512");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "conv1dw: stride/offset alignment");
    MAF_REQUIRE(op->act == MAF_ACT_SILU, "conv1dw: SiLU after both stages (common.py:905-909)");
    C1dArgs a;
    a.x = static_cast<const half_t*>(sr.ptr); a.out = static_cast<half_t*>(op->out);
    a.par = static_cast<const unsigned char*>(op->w);
    a.B = op->B; a.H = op->H; a.W = op->W; a.Cin = op->Cin; a.Cmid = op->Cout; a.S1 = (op->Cin + 31) / 32; a.nMB = (op->Cout + 31) / 32;
    a.x_stride = sr.stride; a.x_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.tilesX = maf_cdiv(a.W, 16); a.tilesY = maf_cdiv(a.H, 16);
    a.nwg = a.B * a.tilesX * a.tilesY * a.nMB;
    a.rec = (int)maf_conv1dw_record_bytes(op->ksize, op->Cin);
    switch (op->ksize) {
        case 3: return launch_k<3>(a, s);
        case 5: return launch_k<5>(a, s);
        case 7: return launch_k<7>(a, s);
        case 9: return launch_k<9>(a, s);
        default: maf_set_error("conv1dw: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
}
