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

    // ---- 2. k (2k) MFMAs per channel set
    const bool toe_active = (p >> 2) == g;
    const int q4 = wave * 4;
    int hi_off = 4;
    asm volatile("" : "+v"(hi_off));                        // keep the two window halves two ds_read_b64 (see bottleneck.hip)
    f32x4_t dacc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) dacc[s] = (f32x4_t)0.f;
    const half8_t* tl = toe + p;
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
