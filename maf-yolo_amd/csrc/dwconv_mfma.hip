// Depth-wise k x k convolution on the matrix cores (k = 7, 9: the 40x40 / 20x20 maps, where the VALU kernel of dwconv.hip is
// bound by its k*k FMAs per output at 0.5-1.1 TB/s).
//
// Replaces the same reference code as dwconv.hip: DilatedReparamBlock.lk_origin after merge_dilated_branches + the folded
// BatchNorms (yolov6/layers/common.py:3025, 3033-3051, 3085-3100), optionally + DepthBottleneckUni.act (:909).
//
// One workgroup = one 16 x 16 output tile x 32 channels:
//   1. the (16+k-1)^2 halo tile is read NHWC (16-byte loads, zero outside the image) and transposed into LDS *planes*
//      [channel][row][x] (ds_write_b16: the price of the layout the next step needs);
//   2. each wave owns 4 output columns; for every tap row ky one MFMA 16x16x32 convolves FOUR channels along x:
//      A = block-diagonal Toeplitz operand (4 outputs x 8-wide window of the filter row, host-built table, staged in LDS),
//      B = the plane rows (16 image rows x the 8-wide window: two 8-byte LDS reads per lane); k instructions (2k for k > 5)
//      accumulate [4 channels][4 x][16 rows] — see bottleneck.hip for the operand algebra and the bank layout;
//   3. after the 8 channel sets a lane holds 8 consecutive channels of 4 pixels: bias + activation, 16-byte NHWC stores.
#include <type_traits>
#include <utility>
#include "maf_common.h"

namespace {

struct DwmArgs {
    const half_t* in; half_t* out;
    const half8_t* toe;   // [nCB][8][K][PARTS][16]  Toeplitz rows, entry (G*4 + r): filter row of channel cb*32 + 8G + s
    const float* bias;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff, act;
    int tilesX, tilesY, nCB, nwg;
};

typedef half_t half4v_t __attribute__((ext_vector_type(4)));

template <int N, int I = 0, typename F>
__device__ __forceinline__ void dwm_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dwm_static_for<N, I + 1>(f);
    }
}
// LDS reads as inline assembly: the caller counts them and waits with dwm_wait_lgkm
template <int OFF> __device__ __forceinline__ void dwm_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void dwm_ds_read_b64(u32x2_t& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void dwm_wait_lgkm(u32x4_t& a, u32x2_t& b, u32x2_t& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }

template <int K>
struct DwmCfg {
    static constexpr int P = K / 2, PARTS = K > 5 ? 2 : 1;
    static constexpr int RH = 16 + K - 1, RWC = (16 + K - 1 + 3) & ~3, NHP = RH * RWC;
    static constexpr int RWP = 24;
    static constexpr int PSB = ((RH * RWP * 2 - 8 + 255) / 256) * 256 + 8;     // plane stride = 8 (mod 256): conflict-free 8-byte reads
    static constexpr int PS = PSB / 2;
    static constexpr int NTOE = 8 * K * PARTS * 16;
    static constexpr size_t LDS = (size_t)32 * PSB + (size_t)NTOE * 16;
};

template <int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_mfma_kernel(const DwmArgs a) {
    typedef DwmCfg<K> Cf;
    constexpr int P = Cf::P, PARTS = Cf::PARTS, RH = Cf::RH, RWC = Cf::RWC, NHP = Cf::NHP, RWP = Cf::RWP, PS = Cf::PS, NTOE = Cf::NTOE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* T1 = reinterpret_cast<half_t*>(smem_raw);                        // [32 slots][PS]; channel 8g+s of the block lives in slot 4s+g
    half8_t* toe = reinterpret_cast<half8_t*>(smem_raw + 32 * Cf::PSB);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int cb = lid % a.nCB;                              // the channel blocks of a tile run back to back (shared halo lines in L2)
    int tt = lid / a.nCB;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty * 16, x0 = tx * 16;
    const half_t* xin = a.in + (size_t)b * a.H * a.W * a.in_stride + a.in_coff + cb * 32;

    // ---- 1. Toeplitz table of this channel block -> LDS; halo tile -> planes
    for (int i = tid; i < NTOE; i += 256) toe[i] = a.toe[(size_t)cb * NTOE + i];
    for (int idx = tid; idx < NHP * 4; idx += 256) {
        const int ch = idx & 3, pix = idx >> 2;
        const int hr = pix / RWC, hc = pix - hr * RWC;
        const int iy = y0 - P + hr, ix = x0 - P + hc;
        half8_t v = (half8_t)(half_t)0;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && cb * 32 + ch * 8 < a.C)
            v = *reinterpret_cast<const half8_t*>(xin + ((size_t)iy * a.W + ix) * a.in_stride + ch * 8);
        half_t* dst = T1 + (size_t)ch * PS + hr * RWP + hc;  // channel ch*8 + j -> slot 4j + ch
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[(size_t)(4 * j) * PS] = v[j];
    }
    __syncthreads();

    // ---- 2. k (2k) MFMAs per channel set: ONE straight line of K * PARTS * 8 steps (tap row, window, channel set), software-pipelined by
    // hand exactly like phase B of csrc/bottleneck.hip — the three LDS reads of step t + BD are inline assembly issued before the MFMA of
    // step t, with hand-counted `s_waitcnt lgkmcnt(n)` (as a loop of plain loads every MFMA sat behind a full LDS round trip).  All 64 lanes
    // read a Toeplitz entry and the 48 lanes outside the block diagonal AND it to zero.
    const uint32_t toe_mask = (p >> 2) == g ? 0xffffffffu : 0u;
    const int q4 = wave * 4;
    int hi_off = 4;
    asm volatile("" : "+v"(hi_off));                        // keep the two window halves two ds_read_b64 (see bottleneck.hip)
    f32x4_t dacc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) dacc[s] = (f32x4_t)0.f;
    const half8_t* tl = toe + p;
    const half_t* t1l = T1 + (size_t)g * PS + p * RWP + q4;
    const half_t* t1h = t1l + hi_off;
    constexpr int NSTEP = K * PARTS * 8, BD = 3;
    u32x4_t avr[BD + 1];
    u32x2_t blo[BD + 1], bhi[BD + 1];
    const uint32_t a_toe = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)tl;
    const uint32_t a_t1l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)t1l;
    const uint32_t a_t1h = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)t1h;
    auto ld_step = [&](auto idx) {
        constexpr int t = decltype(idx)::value;
        if constexpr (t < NSTEP) {
            constexpr int s = t % 8, part = (t / 8) % PARTS, ky = t / (8 * PARTS);
            constexpr int ot = ((s * K + ky) * PARTS + part) * 256, o = (s * 4 * PS + ky * RWP + part * 4) * 2;
            static_assert(ot < 65536 && o < 65536, "ds offset field");
            dwm_ds_read_b128<ot>(avr[t % (BD + 1)], a_toe);
            dwm_ds_read_b64<o>(blo[t % (BD + 1)], a_t1l);
            dwm_ds_read_b64<o>(bhi[t % (BD + 1)], a_t1h);
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the counter now counts only the reads below
    dwm_static_for<BD>([&](auto idx) { ld_step(idx); });
    dwm_static_for<NSTEP>([&](auto idx) {
        constexpr int t = decltype(idx)::value, s = t % 8, sl = t % (BD + 1);
        ld_step(std::integral_constant<int, t + BD>{});
        constexpr int ahead = (NSTEP - 1 - t) < BD ? (NSTEP - 1 - t) : BD;          // steps whose reads were issued after step t's
        dwm_wait_lgkm<3 * ahead>(avr[sl], blo[sl], bhi[sl]);
        const u32x4_t bw = (u32x4_t){blo[sl][0], blo[sl][1], bhi[sl][0], bhi[sl][1]};
        const u32x4_t am = avr[sl] & toe_mask;
        dacc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, am), __builtin_bit_cast(half8_t, bw), dacc[s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                  // keep the issue order as written
    });

    // ---- 3. lane (g, n = p): channels cb*32 + 8g .. +7 of pixels (row y0 + n, x = x0 + 4q + r)
    const int c0 = cb * 32 + g * 8;
    if (c0 >= a.C) return;
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(a.bias + c0), b1 = *reinterpret_cast<const f32x4_t*>(a.bias + c0 + 4);
    const float bs[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const int oy = y0 + p;
    if (oy >= a.H) return;
    half_t* orow = a.out + ((size_t)b * a.H + oy) * a.W * a.out_stride + a.out_coff + c0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ox = x0 + q4 + r;
        if (ox >= a.W) continue;
        half8_t o;
#pragma unroll
        for (int s = 0; s < 8; ++s) o[s] = (half_t)maf_act<ACT>(dacc[s][r] + bs[s]);
        *reinterpret_cast<half8_t*>(orow + (size_t)ox * a.out_stride) = o;
    }
}

template <int K>
int launch_k(const DwmArgs& a, hipStream_t s) {
    constexpr size_t lds = DwmCfg<K>::LDS;
    static_assert(lds <= 160 * 1024, "dwconv_mfma: LDS budget");
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mfma_kernel<K, MAF_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dwconv_mfma)");
        if (!rc) rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mfma_kernel<K, MAF_ACT_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dwconv_mfma)");
        if (rc) return rc;
        attr = true;
    }
    if (a.act == MAF_ACT_SILU) hipLaunchKernelGGL((dwconv_mfma_kernel<K, MAF_ACT_SILU>), dim3(a.nwg), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dwconv_mfma_kernel<K, MAF_ACT_NONE>), dim3(a.nwg), dim3(256), lds, s, a);
    return maf_check_hip(hipGetLastError(), "dwconv_mfma launch");
}

}  // namespace

int maf_launch_dwconv_mfma(const maf_op_t* op, hipStream_t s) {
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->dtype == MAF_F16, "dwconv (matrix-core variant, tile_p = -1): fp16 only");
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr && op->out && op->bias, "dwconv: one direct source");
    MAF_REQUIRE(op->aux[0], "dwconv (tile_p = -1): aux[0] = Toeplitz table (maf-yolo_amd/pack.py:pack_dw_toeplitz)");
    MAF_REQUIRE(op->Cin == op->Cout && sr.C == op->Cin && op->Cin % 8 == 0, "dwconv: C must be a multiple of 8");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->act == MAF_ACT_NONE || op->act == MAF_ACT_SILU, "dwconv: act must be none or silu");
    DwmArgs a;
    a.in = static_cast<const half_t*>(sr.ptr); a.out = static_cast<half_t*>(op->out);
    a.toe = static_cast<const half8_t*>(op->aux[0]); a.bias = op->bias;
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cin;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff; a.act = op->act;
    a.tilesX = maf_cdiv(a.W, 16); a.tilesY = maf_cdiv(a.H, 16); a.nCB = maf_cdiv(a.C, 32);
    a.nwg = a.B * a.tilesX * a.tilesY * a.nCB;
    switch (op->ksize) {
        case 3: return launch_k<3>(a, s);
        case 5: return launch_k<5>(a, s);
        case 7: return launch_k<7>(a, s);
        case 9: return launch_k<9>(a, s);
        default: maf_set_error("dwconv: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
}
