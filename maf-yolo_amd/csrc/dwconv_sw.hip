// Depth-wise k x k (k in {3,5,7,9}) stride-1 "same" convolution + bias (+SiLU), NHWC fp16, with the WEIGHTS AS SCALAR OPERANDS
// (tile_p = -3 of MAF_OP_DWCONV).
//
// Same reference code as dwconv.hip (merged DilatedReparamBlock / UniRepLKNetBlock of the deploy graph, yolov6/layers/common.py:3024-3051,
// 3085-3100; the head's cls_conv / reg_conv, common.py:1329,1333).  What the two older kernels pay for (knock-out builds, round 3: 22 of the
// 47 us of a k = 9 launch with neither loads nor multiply-adds): a lane there owns one 16-byte CHANNEL group of a pixel strip, so the k weight
// vectors of a kernel row are per-lane data (k more LDS reads and 4 k registers per row), the halo tile goes global -> registers -> LDS with
// index arithmetic per element, and a workgroup lives load -> barrier -> compute -> store at two waves per SIMD, where a wave issues one
// vector instruction per ~7 cycles.
//
// Here a WAVE owns one 8-channel group (16 bytes per pixel) of a TH x TW tile of one image and a lane owns pixels:
//   * the wave's halo plane [(TH + k - 1)][PITCH] x 16 B travels global -> LDS by DMA (global_load_lds, one 16-byte gather per lane and round, no
//     staging registers, out-of-image pixels read a zero page); the plane belongs to the wave alone, so the kernel has NO barrier — a wave waits
//     for its own vmcnt and starts; the waves of a workgroup (consecutive channel groups of the same pixels: together whole cache lines) and the
//     workgroups of a CU overlap each other's gathers;
//   * the k x k x 8 weights of the channel group are the same for all 64 lanes: they are read through the scalar cache (s_load_dwordx4 per tap)
//     and enter v_fma_mix_f32 as its SGPR operand — no LDS traffic, no vector registers;
//   * a lane computes a strip of R = 4 pixels along x: per kernel row R + k - 1 ds_read_b128 at immediate offsets from one row address and
//     4 k x 8 multiply-adds from registers (fp32 accumulation, the same sums as dwconv_tile_kernel in the same order);
//     lanes = consecutive strips of the tile in row-major order, PITCH chosen by the launcher so that the strips of an LDS lane group fall on
//     different 16-byte bank slots;
//   * ~95 VGPRs and 7-13 KB of LDS per wave: 4 waves per SIMD, which is what the vector ALU needs to issue every ~1.7 cycles.
#include "maf_common.h"
#include "lds_pipe.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko): 32 = no multiply-add loop, 64 = no halo gather, 128 = no output stores, 256 = no LDS reads, 512 = one weight vector for all taps
#endif

namespace {

constexpr int R = 4;                 // output pixels per lane strip

struct DwsArgs {
    const half_t* in; const half_t* w; const float* bias; half_t* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff;
    int TH, TW, PITCH, SPR;          // tile rows, columns (multiple of R), LDS row pitch in pixels, strips per tile row
    int tilesX, tilesY, nCG, in_groups, nunits, nwg, plane_slots, rounds;
    int gdy, gdx;                    // 64 / PITCH, 64 % PITCH: the step of a lane's plane slot from one gather round to the next
    uint32_t m_cg, m_tx, m_ty, m_pitch, m_spr;      // ceil(2^32 / d) of the divisors used on the device
};

__device__ __attribute__((aligned(16))) unsigned int g_zero16s[4];

typedef const __attribute__((address_space(4))) u32x4_t* cvec4_t;           // constant address space: uniform addresses become s_load
typedef const __attribute__((address_space(4))) f32x4_t* cf32x4_t;

__device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t d, uint32_t m) { return d > 1 ? __umulhi(n, m) : n; }

// acc[c] += x[c] * w[c] over 8 channels; w in scalar registers (wave-uniform)
__device__ __forceinline__ void smac(float (&acc)[8], const u32x4_t& x, const u32x4_t& w) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q]) : "v"(x[q]), "s"(w[q]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q + 1]) : "v"(x[q]), "s"(w[q]));
    }
}

template <int K, int ACT>
__global__ __launch_bounds__(512) void dwconv_sw_kernel(const DwsArgs a) {
    constexpr int P = K / 2, NX = R + K - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;

    int lid;                                                             // XCD-aware bijective remap (as dwconv.hip)
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int unit = lid * nw + wv;                                      // one wave = one (image, tile, channel group); channel groups fastest
    if (unit >= a.nunits) return;                                        // no barrier below: a wave may leave
    uint32_t t = (uint32_t)unit, q;
    q = fdiv(t, a.nCG, a.m_cg); const int cg = (int)(t - q * a.nCG); t = q;
    q = fdiv(t, a.tilesX, a.m_tx); const int tx = (int)(t - q * a.tilesX); t = q;
    q = fdiv(t, a.tilesY, a.m_ty); const int ty = (int)(t - q * a.tilesY);
    const int b = (int)q;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    const int RH = a.TH + K - 1, RW = a.TW + K - 1;
    unsigned char* plane_p = sw_raw + (size_t)wv * a.plane_slots * 16;
    const uint32_t plane = lp_lds_addr(plane_p);

    if (!(MAF_KO & 64)) {   // ---- gather the halo plane of this channel group: slot (py, px) of round r = lane + 64 r, one 16-byte DMA per lane and round
        const int cgi = cg >= a.in_groups ? cg - a.in_groups : cg;       // two filters per input channel (Cout = 2 Cin): the second half reads the same input
        const half_t* img = a.in + (size_t)b * a.H * a.W * a.in_stride + a.in_coff + cgi * 8;
        int py = (int)fdiv((uint32_t)lane, a.PITCH, a.m_pitch), px = lane - py * a.PITCH;
#pragma unroll 1
        for (int r = 0; r < a.rounds; ++r) {
            const int iy = y0 - P + py, ix = x0 - P + px;
            const bool ok = px < RW && py < RH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const half_t* src = ok ? img + ((size_t)iy * a.W + ix) * a.in_stride : reinterpret_cast<const half_t*>(g_zero16s);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(plane_p + r * 1024), 16, 0, 0);
            px += a.gdx; py += a.gdy;
            if (px >= a.PITCH) { px -= a.PITCH; ++py; }
        }
    }
    const cvec4_t wp = (cvec4_t)(uintptr_t)(a.w + cg * 8);               // tap kk at wp[kk * C / 8]
    const cf32x4_t bp = (cf32x4_t)(uintptr_t)(a.bias + cg * 8);
    const f32x4_t b0 = bp[0], b1 = bp[1];
    const int C8 = a.C >> 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int nstrips = a.TH * a.SPR;
    const uint32_t row_step = (uint32_t)a.PITCH * 16;
#pragma unroll 1
    for (int s0 = 0; s0 < nstrips; s0 += 64) {
        const int s = min(s0 + lane, nstrips - 1);                       // idle lanes of the last pass redo the last strip (reads stay inside the plane) and do not store
        const int y = (int)fdiv((uint32_t)s, a.SPR, a.m_spr), sx = s - y * a.SPR;
        uint32_t row = plane + (uint32_t)(y * a.PITCH + R * sx) * 16;
        float acc[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[r][j] = b0[j]; acc[r][4 + j] = b1[j]; }
        }
        if (!(MAF_KO & 32)) {
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                u32x4_t wk[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) wk[kx] = wp[(MAF_KO & 512) ? 0 : (ky * K + kx) * C8];
                u32x4_t x[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    if (MAF_KO & 256) x[i] = u32x4_t{row, row + (uint32_t)i, row, row};
                    else x[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>((uintptr_t)(row + i * 16));
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int kx = i - r;                             // compile-time after unrolling
                        if (kx >= 0 && kx < K) smac(acc[r], x[i], wk[kx]);
                    }
                }
                row += row_step;
            }
        }
        const int oy = y0 + y, ox0 = x0 + R * sx;
        if (!(MAF_KO & 128) && s0 + lane < nstrips && oy < a.H) {
            half_t* out = a.out + a.out_coff + cg * 8 + ((size_t)((size_t)b * a.H + oy) * a.W + ox0) * a.out_stride;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (ox0 + r < a.W) {
                    half8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)maf_act<ACT>(acc[r][j]);
                    *reinterpret_cast<half8_t*>(out + (size_t)r * a.out_stride) = o;
                }
            }
        }
    }
}

constexpr size_t kMaxLdsSw = 160 * 1024;

// LDS cycles of one ds_read_b128 of the first pass (lane = strip in row-major order) for a row pitch: per lane group of the instruction
// (MI355X_MICROARCH.md, LDS: 4 groups of 16 lanes), the largest number of lanes on one 16-byte slot of the 256-byte bank row.
int sw_conflict_cycles(int pitch, int spr, int nstrips) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int total = 0;
    for (int g = 0; g < 4; ++g) {
        int cnt[16] = {0}, worst = 1;
        for (int i = 0; i < 16; ++i) {
            const int s = groups[g][i] < nstrips ? groups[g][i] : nstrips - 1;
            const int slot = ((s / spr) * pitch + R * (s % spr)) & 15;
            worst = cnt[slot] + 1 > worst ? cnt[slot] + 1 : worst;
            ++cnt[slot];
        }
        total += worst;
    }
    return total;
}

int sw_pitch(int TH, int TW, int K) {
    const int RW = TW + K - 1, spr = TW / R, nstrips = TH * spr;
    int best = RW, bc = 1 << 30;
    for (int p = RW; p < RW + 8; ++p) {
        const int c = sw_conflict_cycles(p, spr, nstrips);
        if (c < bc) { bc = c; best = p; }
    }
    return best;
}

uint32_t magic(int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint32_t)d - 1) / (uint32_t)d) : 0u; }

template <int K, int ACT>
int launch_sw(const DwsArgs& a, int nw, hipStream_t s) {
    const size_t lds = (size_t)nw * a.plane_slots * 16;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_sw_kernel<K, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsSw),
                               "hipFuncSetAttribute(dwconv_sw)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv_sw_kernel<K, ACT>), dim3(a.nwg), dim3(64 * nw), lds, s, a);
    return maf_check_hip(hipGetLastError(), "dwconv_sw launch");
}

}  // namespace

// tile_p = -3: tile_c = tile columns (multiple of 4), tile_k = tile rows * 256 + waves per workgroup (1..8; a wave = one 8-channel group of a tile)
int maf_launch_dwconv_sw(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "dwconv (sw): fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr, "dwconv: one direct source");
    MAF_REQUIRE((op->Cout == op->Cin || op->Cout == 2 * op->Cin) && sr.C == op->Cin && op->Cin % 8 == 0,
                "dwconv: Cout = Cin or 2 Cin (two filters per input channel), Cin a multiple of the 16-byte channel group");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->w && op->bias && op->out, "dwconv: null pointer");
    MAF_REQUIRE(op->act == MAF_ACT_NONE || op->act == MAF_ACT_SILU, "dwconv: act must be none or silu");
    const int k = op->ksize;
    MAF_REQUIRE(k == 3 || k == 5 || k == 7 || k == 9, "dwconv: k must be 3, 5, 7 or 9");
    DwsArgs a;
    a.in = static_cast<const half_t*>(sr.ptr); a.w = static_cast<const half_t*>(op->w); a.bias = op->bias; a.out = static_cast<half_t*>(op->out);
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cout; a.in_groups = op->Cin / 8;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.TW = op->tile_c; a.TH = op->tile_k >> 8;
    const int nw = op->tile_k & 255;
    MAF_REQUIRE(a.TW > 0 && a.TW % R == 0 && a.TH > 0 && nw >= 1 && nw <= 8, "dwconv (sw): tile_c = columns (multiple of 4), tile_k = rows * 256 + waves per workgroup (1..8)");
    a.SPR = a.TW / R;
    a.PITCH = sw_pitch(a.TH, a.TW, k);
    a.plane_slots = maf_cdiv((a.TH + k - 1) * a.PITCH, 64) * 64;
    a.rounds = a.plane_slots / 64;
    MAF_REQUIRE((size_t)nw * a.plane_slots * 16 <= kMaxLdsSw, "dwconv (sw): tile does not fit the LDS");
    a.gdy = 64 / a.PITCH; a.gdx = 64 % a.PITCH;
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCG = a.C / 8;
    a.nunits = a.B * a.tilesY * a.tilesX * a.nCG;
    a.nwg = maf_cdiv(a.nunits, nw);
    a.m_cg = magic(a.nCG); a.m_tx = magic(a.tilesX); a.m_ty = magic(a.tilesY); a.m_pitch = magic(a.PITCH); a.m_spr = magic(a.SPR);
    const bool silu = op->act == MAF_ACT_SILU;
    switch (k) {
        case 3: return silu ? launch_sw<3, MAF_ACT_SILU>(a, nw, s) : launch_sw<3, MAF_ACT_NONE>(a, nw, s);
        case 5: return silu ? launch_sw<5, MAF_ACT_SILU>(a, nw, s) : launch_sw<5, MAF_ACT_NONE>(a, nw, s);
        case 7: return silu ? launch_sw<7, MAF_ACT_SILU>(a, nw, s) : launch_sw<7, MAF_ACT_NONE>(a, nw, s);
        default: return silu ? launch_sw<9, MAF_ACT_SILU>(a, nw, s) : launch_sw<9, MAF_ACT_NONE>(a, nw, s);
    }
}
