// Depth-wise k x k (k in {3,5,7,9}) stride-1 "same" convolution + bias (+SiLU), NHWC fp16, on v_dot2_f32_f16 (tile_p = -2 of MAF_OP_DWCONV).
//
// Same reference code as dwconv.hip (merged DilatedReparamBlock / UniRepLKNetBlock of the deploy graph, yolov6/layers/common.py:3024-3051,
// 3085-3100; the head's cls_conv / reg_conv, common.py:1329,1333).  dwconv_tile_kernel spends one v_fma_mix_f32 per (pixel, tap, channel) and
// is bound by the vector ALU issue rate on every layer (k = 9 on the 20 x 20 maps: 81 instructions per output value; 0.5-1.3 TB/s of HBM
// traffic).  v_dot2_f32_f16 does TWO multiply-adds into one fp32 accumulator — the two products must belong to the same output, i.e. be two
// TAPS of one channel — so the halo tile is staged PAIR-INTERLEAVED: the 16-byte LDS vector (row, pair j, quad q) holds, for the 4 channels
// of quad q, the dword (In[row][2j][c], In[row][2j+1][c]) (one v_perm_b32 per dword while staging: the cost is per input element, the gain per
// tap).  An output column t then needs the pairs covering patch columns t .. t + k - 1: for even t they are aligned and take the weight pairs
// (W0,W1),(W2,W3),..,(W[k-1],0); for odd t the pairs start one column early and take (0,W0),(W1,W2),..,(W[k-2],W[k-1]) — two weight-pair
// sets per kernel row (built once per workgroup in LDS), (k + 1) / 2 dot2 instructions per output value and kernel row instead of k.
//
// A lane = one channel quad x NS vertically adjacent strips of RX = 8 output pixels; per kernel row it reads the 2 * (k + 1) / 2 weight-pair
// vectors once (registers, shared by its NS strips) and (RX + k - 1) / 2 input pair vectors per strip: 12-20 dot2 per LDS read, i.e. the
// vector ALU, not the LDS, stays the limit.  fp32 accumulation as before (same sums in a different order).  Quads of neighbouring lanes are
// swapped pairwise at the end so that every lane stores 16 bytes (8 channels of one pixel).
#include "maf_common.h"

#ifndef MAF_KO
#define MAF_KO 0            // profiling builds (make ko): 32 = no multiply-add loop, 64 = no halo loads (zeros are staged), 128 = no output stores
#endif

namespace {

struct Dw2Args {
    const half_t* in; const half_t* w; const float* bias; half_t* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff;
    int TH, TW, CB;                  // tile: rows (multiple of NS), columns (multiple of RX), channels per block (multiple of 8)
    int tilesX, tilesY, nCB, nwg, in_mod;
};

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__host__ __device__ inline int dw2_pair_stride(int nq) {
    for (int ps = nq; ps < nq + 4; ++ps)
        if ((4 * ps - nq) % 16 == 0) return ps;
    return nq + 1;
}

// n / d for n * d < 2^32 by one v_mul_hi: index decoding by runtime tile dimensions costs ~40 instructions per division otherwise, and a lane of
// this kernel issues an instruction every ~4-7 cycles (knock-out builds: with neither loads nor multiply-adds the first version still took 24 of 47 us)
struct FastDiv {
    uint32_t d, m;
    __device__ explicit FastDiv(int dd) : d((uint32_t)dd), m(dd > 1 ? (uint32_t)((0x100000000ull + (uint32_t)dd - 1) / (uint32_t)dd) : 0u) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d > 1 ? __umulhi(n, m) : n; }
};

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
}

template <int K, int RX, int NS, int ACT>
__global__ __launch_bounds__(256) void dwconv_dot2_kernel(const Dw2Args a) {
    constexpr int P = K / 2, NP = (K + 1) / 2, NV = (RX + K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char d2_raw[];
    u32x4_t* tile = reinterpret_cast<u32x4_t*>(d2_raw);                  // [RH][PWP][NQ] pair vectors (+ pad per pair)

    int lid;                                                             // XCD-aware bijective remap (as dwconv.hip)
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int cb = lid % a.nCB;
    int t = lid / a.nCB;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * a.TH, x0 = tx * a.TW, c0 = cb * a.CB;
    const int cbe = min(a.CB, a.C - c0);
    const int NG = cbe / 8, NQ = 2 * NG;                                 // 16-byte channel groups / quads of this block
    const int RH = a.TH + K - 1, PWP = (a.TW + K - 1) / 2;               // patch rows, pixel pairs per row
    // pair stride in vectors: the NQ lanes of a strip read NQ consecutive 16-byte slots and neighbouring strips start RX / 2 = 4 pairs apart — with
    // 4 * PS = NQ (mod 16) the strips of a 16-lane LDS group tile the 256-byte bank row one after the other (NQ = 8: PS = 10, NQ = 4: PS = 5, NQ = 16: 16)
    const int PS = dw2_pair_stride(NQ);
    u32x4_t* wl = tile + RH * PWP * PS;                                  // [K][2 phases][NP][NQ] weight-pair vectors
    const int tid = threadIdx.x;
    const FastDiv dNG(NG), dPWP(PWP), dNQ(NQ);

    {   // ---- stage the halo tile pair-interleaved
        const half_t* in = a.in + a.in_coff + c0 % a.in_mod;
        const int total = RH * PWP * NG;                                 // (row, pair, 8-channel group): two 16-byte loads -> two quad vectors
        for (int base = tid; base < total; base += 256 * 2) {
            u32x4_t va[2], vb[2];
            int dst[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = base + u * 256;
                const int pp = (int)dNG.div((uint32_t)idx), gi = idx - pp * NG;
                const int ry = (int)dPWP.div((uint32_t)pp), pj = pp - ry * PWP;
                const int iy = y0 - P + ry, ix = x0 - P + 2 * pj;
                va[u] = vb[u] = (u32x4_t){0u, 0u, 0u, 0u};
                dst[u] = idx < total ? (ry * PWP + pj) * PS + 2 * gi : -1;
                if (!(MAF_KO & 64) && idx < total && (unsigned)iy < (unsigned)a.H) {
                    const half_t* rowp = in + ((size_t)((size_t)b * a.H + iy) * a.W) * a.in_stride + gi * 8;
                    if ((unsigned)ix < (unsigned)a.W) va[u] = *reinterpret_cast<const u32x4_t*>(rowp + (size_t)ix * a.in_stride);
                    if ((unsigned)(ix + 1) < (unsigned)a.W) vb[u] = *reinterpret_cast<const u32x4_t*>(rowp + (size_t)(ix + 1) * a.in_stride);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (dst[u] >= 0) {
                    u32x4_t q0, q1;                                       // dword c of a quad = (even pixel's channel, odd pixel's channel)
                    q0[0] = __builtin_amdgcn_perm(vb[u][0], va[u][0], 0x05040100u); q0[1] = __builtin_amdgcn_perm(vb[u][0], va[u][0], 0x07060302u);
                    q0[2] = __builtin_amdgcn_perm(vb[u][1], va[u][1], 0x05040100u); q0[3] = __builtin_amdgcn_perm(vb[u][1], va[u][1], 0x07060302u);
                    q1[0] = __builtin_amdgcn_perm(vb[u][2], va[u][2], 0x05040100u); q1[1] = __builtin_amdgcn_perm(vb[u][2], va[u][2], 0x07060302u);
                    q1[2] = __builtin_amdgcn_perm(vb[u][3], va[u][3], 0x05040100u); q1[3] = __builtin_amdgcn_perm(vb[u][3], va[u][3], 0x07060302u);
                    tile[dst[u]] = q0;
                    tile[dst[u] + 1] = q1;
                }
        }
        // weight pairs: wl[((ky * 2 + phase) * NP + p) * NQ + q] dword c = (W[ky][2p - phase][ch], W[ky][2p - phase + 1][ch]), zero outside 0 .. K-1
        const half_t* w = a.w + c0;                                      // [K*K][C]
        // (all of a lane's loads first, then the interleave + LDS stores: as a load -> use loop every round was an L2 round trip)
        constexpr int WR = (K * 2 * NP * 16 + 255) / 256;                 // rounds at the widest block (NQ = 16)
        u32x2_t wlo[WR], whi[WR];
#pragma unroll
        for (int r0 = 0; r0 < WR; ++r0) {
            const int idx = tid + r0 * 256;
            wlo[r0] = whi[r0] = (u32x2_t){0u, 0u};
            if (idx < K * 2 * NP * NQ) {
                int r = (int)dNQ.div((uint32_t)idx);
                const int q = idx - r * NQ;
                const int p = r % NP; r /= NP;                           // (compile-time divisor)
                const int ph = r & 1, ky = r >> 1;
                const int k0 = 2 * p - ph, k1 = k0 + 1;
                // the quad's 4 channels of tap k0 and of tap k1: one 8-byte load each
                if (k0 >= 0 && k0 < K) wlo[r0] = *reinterpret_cast<const u32x2_t*>(w + (size_t)(ky * K + k0) * a.C + 4 * q);
                if (k1 >= 0 && k1 < K) whi[r0] = *reinterpret_cast<const u32x2_t*>(w + (size_t)(ky * K + k1) * a.C + 4 * q);
            }
        }
#pragma unroll
        for (int r0 = 0; r0 < WR; ++r0) {
            const int idx = tid + r0 * 256;
            if (idx < K * 2 * NP * NQ) {
                u32x4_t v;                                               // interleaved channel by channel: dword c = (tap k0, tap k1) of channel c
                v[0] = __builtin_amdgcn_perm(whi[r0][0], wlo[r0][0], 0x05040100u); v[1] = __builtin_amdgcn_perm(whi[r0][0], wlo[r0][0], 0x07060302u);
                v[2] = __builtin_amdgcn_perm(whi[r0][1], wlo[r0][1], 0x05040100u); v[3] = __builtin_amdgcn_perm(whi[r0][1], wlo[r0][1], 0x07060302u);
                wl[idx] = v;
            }
        }
    }
    __syncthreads();

    const int NSX = a.TW / RX, NSY = a.TH / NS;
    const int items = NSY * NSX * NQ;
    const FastDiv dNSX(NSX);
    for (int it = tid; it < items; it += 256) {
        const int u = (int)dNQ.div((uint32_t)it), q = it - u * NQ;
        const int sy = (int)dNSX.div((uint32_t)u), sx = u - sy * NSX;
        const int ry0 = sy * NS, tc0 = sx * RX;                          // first tile row / column of the lane's strips
        if (y0 + ry0 >= a.H || x0 + tc0 >= a.W) continue;
        float acc[NS][RX][4];
        {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(a.bias + c0 + 4 * q);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < RX; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[s][r][c] = bv[c];
        }
#pragma unroll 1
        for (int ky = 0; ky < ((MAF_KO & 32) ? 1 : K); ++ky) {
            u32x4_t we[NP], wo[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                we[p] = wl[((ky * 2 + 0) * NP + p) * NQ + q];
                wo[p] = wl[((ky * 2 + 1) * NP + p) * NQ + q];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const u32x4_t* row = tile + ((ry0 + s + ky) * PWP + tc0 / 2) * PS + q;
                u32x4_t v[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = row[i * PS];
#pragma unroll
                for (int r = 0; r < RX; ++r)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        // even r: pairs r/2 + p with the even-phase weights; odd r: pairs (r-1)/2 + p (one column early) with the odd-phase weights.
                        // The odd phase's last pair index (r-1)/2 + NP - 1 <= NV - 1 always; the even phase's r/2 + NP - 1 <= NV - 1 likewise.
                        const int j = (r >> 1) + p;
                        if (j < NV) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[s][r][c] = dot2(v[j][c], (r & 1) ? wo[p][c] : we[p][c], acc[s][r][c]);
                        }
                    }
            }
        }
        // ---- activation, quad swap with the neighbour lane (q ^ 1: the other 4 channels of the same 8-channel group), 16-byte stores
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int oy = y0 + ry0 + s;
            uint32_t mine[RX][2];
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                const half2_t h0 = {(half_t)maf_act<ACT>(acc[s][r][0]), (half_t)maf_act<ACT>(acc[s][r][1])};
                const half2_t h1 = {(half_t)maf_act<ACT>(acc[s][r][2]), (half_t)maf_act<ACT>(acc[s][r][3])};
                mine[r][0] = __builtin_bit_cast(uint32_t, h0);
                mine[r][1] = __builtin_bit_cast(uint32_t, h1);
            }
            // lane with even q keeps the even pixels, the odd-q lane the odd pixels: each sends the other half (selects, not indexed registers)
            const bool oddq = q & 1;
#pragma unroll
            for (int r = 0; r < RX; r += 2) {
                const uint32_t k0 = oddq ? mine[r + 1][0] : mine[r][0], k1 = oddq ? mine[r + 1][1] : mine[r][1];
                const uint32_t g0 = oddq ? mine[r][0] : mine[r + 1][0], g1 = oddq ? mine[r][1] : mine[r + 1][1];
                const uint32_t o0 = __shfl_xor(g0, 1), o1 = __shfl_xor(g1, 1);
                const int ox = x0 + tc0 + r + (oddq ? 1 : 0);
                if (oy < a.H && ox < a.W && !((MAF_KO & 128) && o0 != 0x12345678u)) {
                    const u32x4_t vv = oddq ? (u32x4_t){o0, o1, k0, k1} : (u32x4_t){k0, k1, o0, o1};
                    *reinterpret_cast<u32x4_t*>(a.out + a.out_coff + c0 + 8 * (q >> 1) + ((size_t)((size_t)b * a.H + oy) * a.W + ox) * a.out_stride) = vv;
                }
            }
        }
    }
}

constexpr size_t kMaxLds2 = 96 * 1024;

size_t lds_bytes2(int TH, int TW, int CB, int K) {
    const int NQ = CB / 4, NP = (K + 1) / 2;
    return ((size_t)(TH + K - 1) * ((TW + K - 1) / 2) * dw2_pair_stride(NQ) + (size_t)K * 2 * NP * NQ) * 16;
}

template <int K, int ACT>
int launch_k(const Dw2Args& a, int ns, hipStream_t s) {
    const size_t lds = lds_bytes2(a.TH, a.TW, a.CB, K);
    const dim3 g(a.nwg), b(256);
#define MAF_D2(NS_)                                                                                                                      \
    {                                                                                                                                    \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_dot2_kernel<K, 8, NS_, ACT>),               \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds2), "hipFuncSetAttribute(dwconv_dot2)"); \
            if (rc) return rc;                                                                                                           \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((dwconv_dot2_kernel<K, 8, NS_, ACT>), g, b, lds, s, a);                                                       \
    }
    if (ns == 2) MAF_D2(2) else MAF_D2(1)
#undef MAF_D2
    return maf_check_hip(hipGetLastError(), "dwconv_dot2 launch");
}

}  // namespace

// tile_p = -2: tile_c = tile columns (multiple of 8), tile_k = rows * 256 + channels per block (rows even -> two strips per lane)
int maf_launch_dwconv_dot2(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "dwconv (dot2): fp16 only");
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr, "dwconv: one direct source");
    MAF_REQUIRE((op->Cout == op->Cin || op->Cout == 2 * op->Cin) && sr.C == op->Cin && op->Cin % 8 == 0,
                "dwconv: Cout = Cin or 2 Cin (two filters per input channel), Cin a multiple of the 16-byte channel group");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->w && op->bias && op->out, "dwconv: null pointer");
    MAF_REQUIRE(op->act == MAF_ACT_NONE || op->act == MAF_ACT_SILU, "dwconv: act must be none or silu");
    Dw2Args a;
    a.in = static_cast<const half_t*>(sr.ptr); a.w = static_cast<const half_t*>(op->w); a.bias = op->bias; a.out = static_cast<half_t*>(op->out);
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cout; a.in_mod = op->Cin;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.TW = op->tile_c; a.TH = op->tile_k >> 8; a.CB = op->tile_k & 255;
    const int k = op->ksize;
    MAF_REQUIRE(k == 3 || k == 5 || k == 7 || k == 9, "dwconv: k must be 3, 5, 7 or 9");
    MAF_REQUIRE(a.TW > 0 && a.TW % 8 == 0 && a.TH > 0 && a.CB > 0 && a.CB % 8 == 0 && a.CB <= 64, "dwconv (dot2): tile_c = columns (multiple of 8), tile_k = rows * 256 + channels (multiple of 8, <= 64)");
    while (a.in_mod % a.CB) a.CB -= 8;                                  // a channel block never straddles the wrap of the input channels
    MAF_REQUIRE(lds_bytes2(a.TH, a.TW, a.CB, k) <= kMaxLds2, "dwconv (dot2): tile does not fit 96 KiB of LDS");
    const int ns = a.TH % 2 == 0 ? 2 : 1;
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    a.nwg = a.B * a.tilesY * a.tilesX * a.nCB;
    const bool silu = op->act == MAF_ACT_SILU;
    switch (k) {
        case 3: return silu ? launch_k<3, MAF_ACT_SILU>(a, ns, s) : launch_k<3, MAF_ACT_NONE>(a, ns, s);
        case 5: return silu ? launch_k<5, MAF_ACT_SILU>(a, ns, s) : launch_k<5, MAF_ACT_NONE>(a, ns, s);
        case 7: return silu ? launch_k<7, MAF_ACT_SILU>(a, ns, s) : launch_k<7, MAF_ACT_NONE>(a, ns, s);
        default: return silu ? launch_k<9, MAF_ACT_SILU>(a, ns, s) : launch_k<9, MAF_ACT_NONE>(a, ns, s);
    }
}
