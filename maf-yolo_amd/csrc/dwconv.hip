// Depth-wise k x k (k in {3,5,7,9}) stride-1 "same" convolution + bias (+SiLU), NHWC.
//
// Replaces the merged DilatedReparamBlock / UniRepLKNetBlock of the deploy graph
// (yolov6/layers/common.py:3024-3026 deploy branch after merge_dilated_branches :3033-3051 and
// reparameterize :3085-3100) plus DepthBottleneckUni.act (common.py:922) where act = SiLU; the
// heads use act = none (common.py:1329,1333).
//
// HBM-bound stencil (4-40 FLOP/B): no MFMA.  Each workgroup owns a 2-D output tile TH x TW of one
// image and one block of CB channels (<= 128 bytes per pixel = one cache line), so the k-row /
// k-column reuse of the stencil stays inside the workgroup instead of bouncing between the L2s of
// different XCDs:
//   1. the (TH+k-1) x (TW+k-1) input halo tile is staged ONCE into LDS with fully coalesced 16-byte
//      loads (zero outside the image), the k*k weight vectors of the channel block next to it;
//   2. one lane = one 16-byte channel group x a 4-pixel strip along x of one tile row; per kernel
//      row it reads (4+k-1) input vectors + k weight vectors from LDS (consecutive lanes =
//      consecutive channel groups => consecutive 16-byte LDS slots) and does 4*k*N fp32 FMAs from
//      registers;
//   3. bias + activation, 16-byte NHWC stores.
// Pixel stride in LDS is padded by 32 bytes so the strips of a wave land on different bank groups.
// blockIdx is remapped so that consecutive logical tiles (channel blocks of the same pixels, then
// x-neighbours) run on the same XCD and share its L2.
#include "maf_common.h"

namespace {

constexpr int R = 4;                 // output pixels per lane strip

struct DwArgs {
    const void* in; const void* w; const float* bias; void* out;
    int B, H, W, C, in_stride, in_coff, out_stride, out_coff, act;
    int TH, TW, CB;                  // tile: rows, cols (multiple of R), channels per block
    int tilesX, tilesY, nCB, nwg;
    int in_mod;                      // output channel c reads input channel c % in_mod (= C, or C/2 for two filters per input channel)
};

template <typename T> struct Vec;
template <> struct Vec<half_t> {
    static constexpr int N = 8;
    typedef half8_t type;
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4_t type;
};

// acc[j] += v[j] * w[j] over one 16-byte vector, fp32 accumulate.  For f16 this is v_fma_mix_f32 (f16 sources read
// straight from the packed registers, no v_cvt and no fp32 copies => ~100 fewer VGPRs than cvt + v_pk_fma_f32).
__device__ __forceinline__ void vmac(float (&acc)[8], const half8_t& v, const half8_t& w) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, v), b = __builtin_bit_cast(u32x4_t, w);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q]) : "v"(a[q]), "v"(b[q]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q + 1]) : "v"(a[q]), "v"(b[q]));
    }
}
__device__ __forceinline__ void vmac(float (&acc)[4], const f32x4_t& v, const f32x4_t& w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[j], w[j], acc[j]);
}

template <typename T, int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const DwArgs a) {
    constexpr int N = Vec<T>::N;
    constexpr int P = K / 2;
    typedef typename Vec<T>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    vec_t* tile = reinterpret_cast<vec_t*>(smem_raw);

    // XCD-aware bijective remap: logical tile ids are contiguous per XCD (block b runs on XCD b % 8)
    int lid;
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
    const int cbe = min(a.CB, a.C - c0);               // channels in this block
    const int CGB = cbe / N;                           // 16-byte channel groups in this block
    const int PS = a.CB / N + 2;                       // LDS pixel stride in vec units (+32 B pad)
    const int RH = a.TH + K - 1, RW = a.TW + K - 1;
    vec_t* wl = tile + RH * RW * PS;                   // [K*K][CGB]
    const int tid = threadIdx.x;

    {   // ---- stage the halo tile and the weights
        const T* in = static_cast<const T*>(a.in) + a.in_coff + c0 % a.in_mod;
        const int total = RH * RW * CGB;
        for (int base = tid; base < total; base += 256 * 4) {          // 4 independent 16-byte loads in flight per lane
            vec_t v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                const int cgi = idx % CGB, p = idx / CGB;
                const int rx = p % RW, ry = p / RW;
                const int iy = y0 - P + ry, ix = x0 - P + rx;
                v[u] = (vec_t)(T)0;
                dst[u] = idx < total ? p * PS + cgi : -1;
                if (idx < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v[u] = *reinterpret_cast<const vec_t*>(in + ((size_t)((size_t)b * a.H + iy) * a.W + ix) * a.in_stride + cgi * N);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) tile[dst[u]] = v[u];
        }
        const T* w = static_cast<const T*>(a.w) + c0;
        for (int idx = tid; idx < K * K * CGB; idx += 256) {
            const int cgi = idx % CGB, kk = idx / CGB;
            wl[idx] = *reinterpret_cast<const vec_t*>(w + (size_t)kk * a.C + cgi * N);
        }
    }
    __syncthreads();

    const int NSX = a.TW / R;
    const int items = a.TH * NSX * CGB;
    for (int it = tid; it < items; it += 256) {
        const int cgi = it % CGB;
        const int u = it / CGB;
        const int s = u % NSX, ry = u / NSX;
        const int oy = y0 + ry, ox0 = x0 + s * R;
        if (oy >= a.H || ox0 >= a.W) continue;
        float acc[R][N];
        {
            const float* bp = a.bias + c0 + cgi * N;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float bv = bp[j];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][j] = bv;
            }
        }
#pragma unroll 1        // one kernel row live at a time: full unrolling hoists every LDS load and spills (512 regs + scratch)
        for (int ky = 0; ky < K; ++ky) {
            const vec_t* row = tile + ((ry + ky) * RW + s * R) * PS + cgi;
            vec_t wv[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wv[kx] = wl[(ky * K + kx) * CGB + cgi];
#pragma unroll
            for (int i = 0; i < R + K - 1; ++i) {
                const vec_t v = row[i * PS];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int kx = i - r;                 // compile-time after unrolling
                    if (kx >= 0 && kx < K) vmac(acc[r], v, wv[kx]);
                }
            }
        }
        T* out = static_cast<T*>(a.out) + a.out_coff + c0 + cgi * N + ((size_t)((size_t)b * a.H + oy) * a.W + ox0) * a.out_stride;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (ox0 + r < a.W) {
                vec_t o;
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = (T)maf_act<ACT>(acc[r][j]);
                *reinterpret_cast<vec_t*>(out + (size_t)r * a.out_stride) = o;
            }
        }
    }
}

constexpr size_t kMaxLds = 96 * 1024;

size_t lds_bytes(int TH, int TW, int CB, int N, int K) {
    return ((size_t)(TH + K - 1) * (TW + K - 1) * (CB / N + 2) + (size_t)K * K * (CB / N)) * 16;
}

// Pick the tile with the least halo read amplification that fits LDS; prefer full 128-byte channel blocks.
void choose_tile(int H, int W, int C, int N, int K, int& TH, int& TW, int& CB) {
    const int line = 8 * N;                                    // channels per 128-byte line
    double best = 1e30;
    TH = 4; TW = 8; CB = 2 * N;
    const int ths[] = {4, 8, 16, 32}, tws[] = {8, 16, 32};
    for (int cbm = 8; cbm >= 2; cbm >>= 1) {
        const int cb = min(cbm * N, (C + N - 1) / N * N);
        for (int th0 : ths) for (int tw0 : tws) {
            const int th = min(th0, H), tw = min(tw0, (W + R - 1) / R * R);
            const size_t lds = lds_bytes(th, tw, cb, N, K);
            if (lds > kMaxLds) continue;
            const double rows = min(th + K - 1, H), cols = min(tw + K - 1, W);
            double cost = rows * cols / ((double)min(th, H) * min(tw, W));
            if (cb < line && cb < C) cost *= 1.25;             // partial-line loads
            if (lds > 64 * 1024) cost *= 1.5;                  // one workgroup per CU: staging latency is exposed
            else if (lds > 40 * 1024) cost *= 1.25;            // <= 2-3 workgroups per CU
            else if (lds > 20 * 1024) cost *= 1.1;
            const int items = th * (tw / R) * (cb / N);
            if (items < 256) cost *= 256.0 / items;            // idle lanes
            if (cost < best - 1e-9) { best = cost; TH = th; TW = tw; CB = cb; }
        }
    }
}

template <typename T, int ACT>
int launch_t(DwArgs& a, int k, hipStream_t s) {
    constexpr int N = Vec<T>::N;
    if (a.TH > 0 && a.TW > 0 && a.CB > 0) {                       // caller-chosen tile (Plan.autotune): validate it
        if (a.TW % R != 0 || a.CB % N != 0 || a.CB > 8 * N || lds_bytes(a.TH, a.TW, a.CB, N, k) > kMaxLds) {
            maf_set_error("dwconv: bad tile (tile_c multiple of 4, tile_k multiple of the channel group and <= 128 B, LDS <= 96 KiB)");
            return MAF_E_ARG;
        }
    } else {
        choose_tile(a.H, a.W, a.C, N, k, a.TH, a.TW, a.CB);
    }
    while (a.in_mod % a.CB) a.CB -= N;                                  // a channel block never straddles the wrap of the input channels
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    a.nwg = a.B * a.tilesY * a.tilesX * a.nCB;
    const size_t lds = lds_bytes(a.TH, a.TW, a.CB, N, k);
    const dim3 g(a.nwg), b(256);
#define MAF_DW(KK)                                                                                                  \
    {                                                                                                               \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_tile_kernel<T, KK, ACT>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds),   \
                                   "hipFuncSetAttribute(dwconv)");                                                  \
            if (rc) return rc;                                                                                      \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((dwconv_tile_kernel<T, KK, ACT>), g, b, lds, s, a);                                      \
    }
    switch (k) {
        case 3: MAF_DW(3) break;
        case 5: MAF_DW(5) break;
        case 7: MAF_DW(7) break;
        case 9: MAF_DW(9) break;
        default: maf_set_error("dwconv: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
#undef MAF_DW
    return maf_check_hip(hipGetLastError(), "dwconv launch");
}

}  // namespace

int maf_launch_dwconv_mfma(const maf_op_t* op, hipStream_t s);
int maf_launch_dwconv_dot2(const maf_op_t* op, hipStream_t s);
int maf_launch_dwconv_p2(const maf_op_t* op, hipStream_t s);

int maf_launch_dwconv(const maf_op_t* op, hipStream_t s) {
    if (op->tile_p == -1) return maf_launch_dwconv_mfma(op, s);          // matrix-core variant (dwconv_mfma.hip)
    if (op->tile_p == -2) return maf_launch_dwconv_dot2(op, s);          // v_dot2_f32_f16 over tap pairs (dwconv_dot2.hip)
    if (op->tile_p == -4) return maf_launch_dwconv_p2(op, s);            // v_dot2c with scalar weight pairs over an input stored as pixel pairs (dwconv_p2.hip)
    MAF_REQUIRE(op->nsrc < 1 || op->src[0].mode != MAF_SRC_PAIRS, "dwconv: a pixel-pair source needs tile_p = -4");
    MAF_REQUIRE(op->tile_p != -3, "dwconv: tile_p = -3 (the scalar-weight variant) left the library in round 6 — no tuning file of three rounds ever picked it");
    MAF_REQUIRE(op->dtype == MAF_F16 || op->dtype == MAF_F32, "dwconv: dtype must be f16/f32");
    const int N = op->dtype == MAF_F16 ? 8 : 4;
    const maf_src_t& sr = op->src[0];
    MAF_REQUIRE(op->nsrc == 1 && sr.mode == MAF_SRC_DIRECT && sr.ptr, "dwconv: one direct source");
    MAF_REQUIRE((op->Cout == op->Cin || op->Cout == 2 * op->Cin) && sr.C == op->Cin && op->Cin % N == 0,
                "dwconv: Cout = Cin or 2 Cin (two filters per input channel), Cin a multiple of the 16-byte channel group");
    MAF_REQUIRE(sr.stride % N == 0 && sr.coff % N == 0 && op->out_stride % N == 0 && op->out_coff % N == 0, "dwconv: strides/offsets must be 16-byte aligned");
    MAF_REQUIRE(op->w && op->bias && op->out, "dwconv: null pointer");
    DwArgs a;
    a.in = sr.ptr; a.w = op->w; a.bias = op->bias; a.out = op->out;
    a.B = op->B; a.H = op->H; a.W = op->W; a.C = op->Cout; a.in_mod = op->Cin;
    a.in_stride = sr.stride; a.in_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.act = op->act;
    a.TH = op->tile_p; a.TW = op->tile_c; a.CB = op->tile_k;     // 0 = cost-model choice
    MAF_REQUIRE(op->act == MAF_ACT_NONE || op->act == MAF_ACT_SILU, "dwconv: act must be none or silu");
    if (op->dtype == MAF_F16) return op->act == MAF_ACT_SILU ? launch_t<half_t, MAF_ACT_SILU>(a, op->ksize, s) : launch_t<half_t, MAF_ACT_NONE>(a, op->ksize, s);
    return op->act == MAF_ACT_SILU ? launch_t<float, MAF_ACT_SILU>(a, op->ksize, s) : launch_t<float, MAF_ACT_NONE>(a, op->ksize, s);
}
