// Depth-wise k x k (k in {3,5,7,9}) stride-1 "same" convolution + bias (+SiLU), fp16, on v_dot2c_f32_f16 with scalar weight operands over an
// input stored as PIXEL PAIRS (src mode MAF_SRC_PAIRS, tile_p = -4 of MAF_OP_DWCONV).
//
// Same reference code as dwconv.hip (merged DilatedReparamBlock / UniRepLKNetBlock of the deploy graph, yolov6/layers/common.py:3024-3051,
// 3085-3100; the head's cls_conv / reg_conv, common.py:1329,1333).
//
// The depth-wise layers are bound by the vector ALU: a wave64 multiply-add instruction occupies its SIMD for 4 cycles whatever its kind
// (tools/valu_probe.py against the wall clock: 1.9 ns per v_fma_mix_f32 / v_dot2c_f32_f16 / v_pk_fma_f16 with 8 waves per SIMD, 2.06 with 4,
// 2.5 with 2, 3.4 with one), so the k x k x (pixels x channels / 64) instructions of v_fma_mix_f32 alone are 147 us over the nine launches of
// MAF-YOLO-n at bs 32.  v_dot2c does two taps per instruction if the two taps of ONE channel sit in one dword — which NHWC never gives; the
// older dot2 kernel (dwconv_dot2.hip) pays a v_perm per staged dword plus the staging through registers for it and gained 5-15 %.  Here the
// PRODUCER (the 1x1 conv in front of every depth-wise conv of the graph, csrc/conv_stream_lds.inc.h: a lane's accumulators are 4 consecutive
// pixels of its channels) stores the tensor as pixel pairs — [B][H][W / 2][C][2] halfs: the dword (x[2q][c], x[2q+1][c]) — and this kernel needs
// no data movement instruction at all:
//   * a WAVE owns 8 channels of a TH x TW tile: two 16-byte planes (4 channels x 2 pixels per slot) [(TH + k - 1)][PITCH pairs], gathered
//     global -> LDS by DMA (zero page outside the image); planes are private to the wave: no barrier in the kernel;
//   * a lane owns a strip of 4 pixels (two pairs) of a tile row; per kernel row it reads the 2 x (P + 2 or 3) pair slots that cover its window and
//     issues 4 x 8 x (k + 1) / 2 v_dot2c with the weight pairs as SCALAR operands: output pixel t of the strip, window starting at pair
//     relative pixel t + e (e = P & 1), takes the pairs j = m + ((t + e) >> 1) with the EVEN weight set (w[2m], w[2m+1]) if t + e is even and
//     the ODD set (w[2m-1], w[2m]) otherwise (taps outside the kernel are zero weights) — (k + 1) / 2 instructions per kernel row and output
//     value instead of k, fp32 accumulation as everywhere;
//   * the weights of a channel group and kernel row are 16 (k + 1) / 2 dwords, packed on the host in exactly the order the row consumes them
//     (maf-yolo_amd/pack.py:pack_dw_pairs) and read through the scalar cache;
//   * two filters per input channel (the head's cls / reg pair) reuse the wave's planes: one gather, two passes over the weights.
#include "maf_common.h"
#include "lds_pipe.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko): 32 = no multiply-add loop, 64 = no halo gather, 128 = no output stores, 256 = no LDS reads, 512 = one weight row for all
#endif

namespace {

constexpr int R = 4;                 // output pixels per lane strip (two pairs)

struct Dp2Args {
    const half_t* in; const uint32_t* w; const float* bias; half_t* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff;
    int TH, TW, PITCH, SPR;          // tile rows, columns (multiple of 4), LDS row pitch in pairs, strips per tile row
    int tilesX, tilesY, in_groups, nf, nunits, nwg, sub_slots, rounds;
    uint32_t m_ig, m_tx, m_ty, m_pitch, m_spr, m_sub;   // ceil(2^32 / d) of the divisors used on the device
};

__device__ __attribute__((aligned(16))) unsigned int g_zero16p[4];

typedef const __attribute__((address_space(4))) u32x4_t* cvec4_t;           // constant address space: uniform addresses become s_load
typedef const __attribute__((address_space(4))) f32x4_t* cf32x4_t;

__device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t d, uint32_t m) { return d > 1 ? __umulhi(n, m) : n; }

// acc[c] += x[c].lo * w[c].lo + x[c].hi * w[c].hi over 4 channels; w in scalar registers (wave-uniform)
__device__ __forceinline__ void sdot(float* acc, const u32x4_t& x, const u32x4_t& w) {
#pragma unroll
    for (int d = 0; d < 4; ++d) asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[d]) : "s"(w[d]), "v"(x[d]));
}

// NF = 0: every lane stores its own 16-byte pieces (a wave's store instruction touches 64 different lines).  NF = 1 / 2 (filters per input channel;
// launch-time choice, see maf_launch_dwconv_p2): STAGED stores — the nw waves of a workgroup hold nw ADJACENT channel groups of one tile (channel groups
// are the fastest unit index and in_groups % nw == 0), so once every wave has finished with its planes the results go through the dead planes
// ([strip pixel r][wave][strip], 65 slots per row: conflict-free both ways) and leave as nw x 16-byte RUNS per pixel: 16 runs of 64 bytes per store
// instruction (nw = 4) instead of 64 scattered 16-byte pieces.  The results of a filter pass wait in registers (8 per filter) for the last pass.
template <int K, int ACT, int NF>
__global__ __launch_bounds__(512) void dwconv_p2_kernel(const Dp2Args a) {
    constexpr bool STG = NF > 0;
    constexpr int NSP = 65;
    constexpr int P = K / 2, E = P & 1, PE = P + E, NP = (K + 1) / 2, NPL = (R + 2 * PE) / 2 + (E ? 0 : 0);
    static_assert(NPL == (E ? P + 3 : P + 2), "pairs per window");
    extern __shared__ __attribute__((aligned(16))) unsigned char p2_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;

    int lid;                                                             // XCD-aware bijective remap (as dwconv.hip)
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int unit = lid * nw + wv;                                      // one wave = one (image, tile, input channel group); channel groups fastest
    if (unit >= a.nunits) return;                                        // no barrier below unless STG (then nunits % nw == 0: nobody leaves)
    uint32_t t_ = (uint32_t)unit, q_;
    q_ = fdiv(t_, a.in_groups, a.m_ig); const int cgi = (int)(t_ - q_ * a.in_groups); t_ = q_;
    q_ = fdiv(t_, a.tilesX, a.m_tx); const int tx = (int)(t_ - q_ * a.tilesX); t_ = q_;
    q_ = fdiv(t_, a.tilesY, a.m_ty); const int ty = (int)(t_ - q_ * a.tilesY);
    const int b = (int)q_;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    const int RH = a.TH + K - 1, RWP = a.TW / 2 + PE;
    const int wave_slots = a.rounds * 64;
    unsigned char* plane_p = p2_raw + (size_t)wv * wave_slots * 16;
    const uint32_t plane = lp_lds_addr(plane_p);

    if (!(MAF_KO & 64)) {   // ---- gather: slot v = 64 r + lane -> (plane h, row py, pair pp); one 16-byte DMA per lane and round
        const half_t* img = a.in + (size_t)b * a.H * a.W * a.in_stride + (size_t)(a.in_coff + cgi * 8) * 2;
        const int pq0 = (x0 - PE) >> 1;                                  // first pair of the window (x0 even, PE even; may be negative)
#pragma unroll 1
        for (int r = 0; r < a.rounds; ++r) {
            const uint32_t v = (uint32_t)(r * 64 + lane);
            const uint32_t h = v >= (uint32_t)a.sub_slots ? 1u : 0u;
            const uint32_t u = v - h * (uint32_t)a.sub_slots;
            const int py = (int)fdiv(u, a.PITCH, a.m_pitch), pp = (int)u - py * a.PITCH;
            const int iy = y0 - P + py, iq = pq0 + pp;
            const bool ok = pp < RWP && py < RH && (unsigned)iy < (unsigned)a.H && (unsigned)iq < (unsigned)(a.W >> 1);
            const half_t* src = ok ? img + ((size_t)iy * a.W + 2 * iq) * a.in_stride + h * 8 : reinterpret_cast<const half_t*>(g_zero16p);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(plane_p + r * 1024), 16, 0, 0);
        }
    }
    const int nstrips = a.TH * a.SPR;
    const uint32_t row_step = (uint32_t)a.PITCH * 16, sub_bytes = (uint32_t)a.sub_slots * 16;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    half8_t hold[STG ? NF : 1][R];
    int hold_s = 0;
    lp_static_for<(STG ? NF : 1)>([&](auto fidx) {
    constexpr int FI = decltype(fidx)::value;
#pragma unroll 1
    for (int f = FI; f < (STG ? FI + 1 : a.nf); ++f) {
        const int cg = cgi + f * a.in_groups;
        const cvec4_t wp = (cvec4_t)(uintptr_t)(a.w + (size_t)cg * (K * 16 * NP));       // row ky at wp[ky * 4 NP], entry ((h * 2 + set) * NP + m)
        const cf32x4_t bp = (cf32x4_t)(uintptr_t)(a.bias + cg * 8);
        const f32x4_t b0 = bp[0], b1 = bp[1];
#pragma unroll 1
        for (int s0 = 0; s0 < nstrips; s0 += 64) {
            const int s = min(s0 + lane, nstrips - 1);                   // idle lanes of the last pass redo the last strip (reads stay inside the planes) and do not store
            const int y = (int)fdiv((uint32_t)s, a.SPR, a.m_spr), sx = s - y * a.SPR;
            uint32_t row = plane + (uint32_t)(y * a.PITCH + 2 * sx) * 16;
            float acc[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[r][j] = b0[j]; acc[r][4 + j] = b1[j]; }
            }
            if (!(MAF_KO & 32)) {
                // 2 K steps (kernel row ky, channel half h), ONE straight line, software-pipelined: the weight pairs (scalar loads) and the pair slots
                // (LDS reads) of step s + 1 are issued before the multiply-adds of step s, so that a wave waits for operands it asked for a whole
                // step (80-ish instructions) ago — with the loads at the top of their own step every step began with an exposed scalar-cache /
                // LDS round trip (knock-out builds: 5 of 29 us on a k = 9 layer, 10 of 55 on the 80 x 80 head).
                constexpr int NS = 2 * K;
                u32x4_t wk[2][2][NP], x[2][NPL];
                auto load_step = [&](auto idx) {
                    constexpr int s_ = decltype(idx)::value, bf = s_ & 1, ky = s_ >> 1, h = s_ & 1;
                    const cvec4_t wr = wp + ((MAF_KO & 512) ? 0 : s_ * 2 * NP);
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
#pragma unroll
                        for (int m = 0; m < NP; ++m) wk[bf][st][m] = wr[st * NP + m];
                    }
                    const uint32_t ad = row + (uint32_t)ky * row_step + (uint32_t)h * sub_bytes;
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        if (MAF_KO & 256) x[bf][j] = u32x4_t{ad, ad + (uint32_t)j, ad, ad};
                        else x[bf][j] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>((uintptr_t)(ad + j * 16));
                    }
                };
                load_step(std::integral_constant<int, 0>{});
                lp_static_for<NS>([&](auto idx) {
                    constexpr int s_ = decltype(idx)::value, bf = s_ & 1, h = s_ & 1;
                    if constexpr (s_ + 1 < NS) load_step(std::integral_constant<int, s_ + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < NP; ++m) {
#pragma unroll
                        for (int t = 0; t < R; ++t) sdot(&acc[t][4 * h], x[bf][m + ((t + E) >> 1)], wk[bf][(t + E) & 1][m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            const int oy = y0 + y, ox0 = x0 + R * sx;
            if constexpr (STG) {                                         // (one pass: nstrips <= 64)
                hold_s = s;
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) hold[FI][r][j] = (half_t)maf_act<ACT>(acc[r][j]);
                }
            } else
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
    });
    if constexpr (STG) {
        half8_t* stg = reinterpret_cast<half8_t*>(p2_raw);               // [R][nw][NSP] 16-byte slots in the workgroup's plane area
        const int lg = 31 - __builtin_clz((unsigned)nw);                 // nw is a power of two
        const int cg0 = cgi - wv;                                        // the workgroup's first channel group (cgi % nw == wv)
        const int npieces = nstrips * R * nw;
        half_t* obase = a.out + a.out_coff + (size_t)b * a.H * a.W * a.out_stride;
        lp_static_for<NF>([&](auto fidx) {
            constexpr int FI = decltype(fidx)::value;
            __syncthreads();                                             // every wave has left its planes (FI = 0) / read the previous filter's runs back
            if (lane < nstrips) {
#pragma unroll
                for (int r = 0; r < R; ++r) stg[(r * nw + wv) * NSP + hold_s] = hold[FI][r];
            }
            __syncthreads();
            if (!(MAF_KO & 128)) {
                for (int q = (int)threadIdx.x; q < npieces; q += (int)blockDim.x) {
                    const int w = q & (nw - 1), pr = q >> lg, r = pr & (R - 1), s = pr >> 2;
                    const int y = (int)fdiv((uint32_t)s, a.SPR, a.m_spr), sx = s - y * a.SPR;
                    const int oy = y0 + y, ox = x0 + R * sx + r;
                    if (oy < a.H && ox < a.W)
                        *reinterpret_cast<half8_t*>(obase + ((size_t)oy * a.W + ox) * a.out_stride + (size_t)(cg0 + w + FI * a.in_groups) * 8) = stg[(r * nw + w) * NSP + s];
                }
            }
        });
    }
}

constexpr size_t kMaxLdsP2 = 160 * 1024;

// LDS cycles of one ds_read_b128 of the first pass (lane = strip in row-major order, strips two slots apart) for a row pitch: per lane group of
// the instruction (MI355X_MICROARCH.md, LDS: 4 groups of 16 lanes), the largest number of lanes on one 16-byte slot of the 256-byte bank row.
int p2_conflict_cycles(int pitch, int spr, int nstrips) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int total = 0;
    for (int g = 0; g < 4; ++g) {
        int cnt[16] = {0}, worst = 1;
        for (int i = 0; i < 16; ++i) {
            const int s = groups[g][i] < nstrips ? groups[g][i] : nstrips - 1;
            const int slot = ((s / spr) * pitch + 2 * (s % spr)) & 15;
            worst = cnt[slot] + 1 > worst ? cnt[slot] + 1 : worst;
            ++cnt[slot];
        }
        total += worst;
    }
    return total;
}

int p2_pitch(int TH, int TW, int K) {
    const int P = K / 2, PE = P + (P & 1), RWP = TW / 2 + PE, spr = TW / R, nstrips = TH * spr;
    int best = RWP, bc = 1 << 30;
    for (int p = RWP; p < RWP + 8; ++p) {
        const int c = p2_conflict_cycles(p, spr, nstrips);
        if (c < bc) { bc = c; best = p; }
    }
    return best;
}

uint32_t magic(int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint32_t)d - 1) / (uint32_t)d) : 0u; }

template <int K, int ACT, int NF>
int launch_p2_nf(const Dp2Args& a, int nw, hipStream_t s) {
    const size_t lds = (size_t)nw * a.rounds * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_p2_kernel<K, ACT, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsP2),
                               "hipFuncSetAttribute(dwconv_p2)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv_p2_kernel<K, ACT, NF>), dim3(a.nwg), dim3(64 * nw), lds, s, a);
    return maf_check_hip(hipGetLastError(), "dwconv_p2 launch");
}

template <int K, int ACT>
int launch_p2(const Dp2Args& a, int nw, bool stage, hipStream_t s) {
    if (stage) return a.nf == 2 ? launch_p2_nf<K, ACT, 2>(a, nw, s) : launch_p2_nf<K, ACT, 1>(a, nw, s);
    return launch_p2_nf<K, ACT, 0>(a, nw, s);
}

}  // namespace

// tile_p = -4: src[0].mode = MAF_SRC_PAIRS; aux[1] = weight pairs (pack.py:pack_dw_pairs); tile_c = tile columns (multiple of 4),
// tile_k = tile rows * 256 + waves per workgroup (1..8; a wave = 8 input channels of a tile) [+ 128: staged stores — the workgroup's waves must then be
// adjacent channel groups of one tile (2, 4 or 8 waves dividing Cin / 8), a filter pass one round of strips (rows * columns / 4 <= 64) and the planes
// of a wave hold 4 x 65 slots]
int maf_launch_dwconv_p2(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "dwconv (pairs): fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_PAIRS && sr.ptr, "dwconv (pairs): one source stored as pixel pairs (MAF_SRC_PAIRS)");
    MAF_REQUIRE((op->Cout == op->Cin || op->Cout == 2 * op->Cin) && sr.C == op->Cin && op->Cin % 8 == 0,
                "dwconv: Cout = Cin or 2 Cin (two filters per input channel), Cin a multiple of the 16-byte channel group");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 4 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->W % 2 == 0, "dwconv (pairs): even width");
    MAF_REQUIRE(op->aux[1] && op->bias && op->out, "dwconv (pairs): null pointer (aux[1] = weight pairs)");
    MAF_REQUIRE(op->act == MAF_ACT_NONE || op->act == MAF_ACT_SILU, "dwconv: act must be none or silu");
    const int k = op->ksize;
    MAF_REQUIRE(k == 3 || k == 5 || k == 7 || k == 9, "dwconv: k must be 3, 5, 7 or 9");
    Dp2Args a;
    a.in = static_cast<const half_t*>(sr.ptr); a.w = static_cast<const uint32_t*>(op->aux[1]); a.bias = op->bias; a.out = static_cast<half_t*>(op->out);
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cout; a.in_groups = op->Cin / 8; a.nf = op->Cout / op->Cin;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.TW = op->tile_c; a.TH = op->tile_k >> 8;
    const int nw = op->tile_k & 127;
    const bool stage = (op->tile_k & 128) != 0;
    MAF_REQUIRE(a.TW > 0 && a.TW % R == 0 && a.TH > 0 && nw >= 1 && nw <= 8, "dwconv (pairs): tile_c = columns (multiple of 4), tile_k = rows * 256 + waves per workgroup (1..8) [+ 128]");
    a.SPR = a.TW / R;
    a.PITCH = p2_pitch(a.TH, a.TW, k);
    a.sub_slots = (a.TH + k - 1) * a.PITCH;
    a.rounds = maf_cdiv(2 * a.sub_slots, 64);
    MAF_REQUIRE((size_t)nw * a.rounds * 1024 <= kMaxLdsP2, "dwconv (pairs): tile does not fit the LDS");
    MAF_REQUIRE(!stage || (nw >= 2 && (nw & (nw - 1)) == 0 && a.in_groups % nw == 0 && a.TH * a.SPR <= 64 && a.rounds * 1024 >= R * 65 * 16),
                "dwconv (pairs, staged stores): 2 / 4 / 8 waves dividing Cin / 8, rows * columns / 4 <= 64, planes of >= 4160 bytes per wave");
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH);
    a.nunits = a.B * a.tilesY * a.tilesX * a.in_groups;
    a.nwg = maf_cdiv(a.nunits, nw);
    a.m_ig = magic(a.in_groups); a.m_tx = magic(a.tilesX); a.m_ty = magic(a.tilesY); a.m_pitch = magic(a.PITCH); a.m_spr = magic(a.SPR); a.m_sub = magic(a.sub_slots);
    const bool silu = op->act == MAF_ACT_SILU;
    switch (k) {
        case 3: return silu ? launch_p2<3, MAF_ACT_SILU>(a, nw, stage, s) : launch_p2<3, MAF_ACT_NONE>(a, nw, stage, s);
        case 5: return silu ? launch_p2<5, MAF_ACT_SILU>(a, nw, stage, s) : launch_p2<5, MAF_ACT_NONE>(a, nw, stage, s);
        case 7: return silu ? launch_p2<7, MAF_ACT_SILU>(a, nw, stage, s) : launch_p2<7, MAF_ACT_NONE>(a, nw, stage, s);
        default: return silu ? launch_p2<9, MAF_ACT_SILU>(a, nw, stage, s) : launch_p2<9, MAF_ACT_NONE>(a, nw, stage, s);
    }
}
