// The parallel depth-wise branches of a DilatedReparamBlock in TRAIN form as ONE launch per direction (training only; the deploy graph has the
// branches merged into one k x k filter, dwconv*.hip).
//
// Reference: yolov6/layers/common.py:3024-3031 — out = origin_bn(lk_origin(x)) + sum_j dil_bn_j(dil_conv_j(x)): NB depth-wise convolutions of the
// SAME input with kernel sizes K0, K0 - 2, ... (K0 = 3: 3, 3, 1; K0 = 5: 5, 3, 1 — the 1 x 1 branch, a per-channel scale, rides along), each
// followed by its own BatchNorm.  As separate launches (round 2 / 3) a step of MAF-YOLO-n ran 46 depth-wise forward kernels that read x 2 - 4
// times per block, 46 data-gradient kernels that each wrote a full dx_j, and 27 element-wise adds that summed them:
//   forward   x is staged ONCE (halo of K0), the NB filters walk the same LDS tile, NB outputs                (1 + NB passes instead of 2 NB)
//   dgrad     dx = sum_j corr(dz_j, flip(w_j)): the branches' dz_j are staged one after the other, the sum stays in registers (fp32) and is
//             written once                                                                                    (NB + 1 passes instead of 5 NB - 3)
// Same tile scheme as dwconv.hip (2-D tile x channel block per workgroup, 4-pixel strips per lane, v_fma_mix_f32); the weight gradients stay
// per branch (train_ops.hip: dw_wgrad_kernel on the weight-gradient stream).
#include "maf_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int R = 4;                 // output pixels per lane strip
constexpr int MAXB = 4;

struct DwbArgs {
    const void* src[MAXB]; void* dst[MAXB]; const void* w[MAXB];
    int src_stride[MAXB], dst_stride[MAXB];
    int B, H, W, C, TH, TW, CB, tilesX, tilesY, nCB, nwg;
    float* stats[MAXB]; int stats_R;         // forward only: per branch {sum, sum of squares} of the stored outputs -> [stats_R][2][C] (the BatchNorm's scratch half), or null
};

template <typename T> struct Vec;
template <> struct Vec<half_t> { static constexpr int N = 8; typedef half8_t type; };
template <> struct Vec<float> { static constexpr int N = 4; typedef f32x4_t type; };

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

template <int N, int I = 0, typename F>
__device__ __forceinline__ void db_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        db_static_for<N, I + 1>(f);
    }
}

// kernel size of branch J of a block whose large kernel is K0 (common.py:2997-3008): 3 -> 3, 3, 1;  5 -> 5, 3, 1;  7 -> 7, 5, 3;  9 -> 9, 7, 5, 3
template <int K0, int J> constexpr int branch_k() { return K0 == 3 ? (J < 2 ? 3 : 1) : K0 - 2 * J; }

// acc[r][:] += sum over the K x K window of the strip's 4 pixels; `row0` = LDS vector of the window's top-left pixel for strip pixel 0, `rw` = tile row pitch in pixels
template <typename T, int K>
__device__ __forceinline__ void strip_mac(float (&acc)[R][Vec<T>::N], const typename Vec<T>::type* row0, int rw, int PS, const typename Vec<T>::type* wl, int CGB, int cgi) {
    typedef typename Vec<T>::type vec_t;
#pragma unroll 1        // one kernel row live at a time (as dwconv.hip: full unrolling hoists every LDS load and spills)
    for (int ky = 0; ky < K; ++ky) {
        const vec_t* row = row0 + ky * rw * PS;
        vec_t wv[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wv[kx] = wl[(ky * K + kx) * CGB + cgi];
#pragma unroll
        for (int i = 0; i < R + K - 1; ++i) {
            const vec_t v = row[i * PS];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kx = i - r;
                if (kx >= 0 && kx < K) vmac(acc[r], v, wv[kx]);
            }
        }
    }
}

struct TileId { int b, y0, x0, c0, cbe; };

__device__ __forceinline__ TileId decode_tile(const DwbArgs& a) {
    int lid;                                                    // XCD-aware bijective remap (as dwconv.hip): logical tiles contiguous per XCD
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.nwg >> 3, r = a.nwg & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    TileId t;
    const int cb = lid % a.nCB;
    int u = lid / a.nCB;
    const int tx = u % a.tilesX; u /= a.tilesX;
    const int ty = u % a.tilesY;
    t.b = u / a.tilesY; t.y0 = ty * a.TH; t.x0 = tx * a.TW; t.c0 = cb * a.CB; t.cbe = min(a.CB, a.C - t.c0);
    return t;
}

// halo tile of `in` (window of K around the TH x TW tile, zeros outside the image) -> LDS [RH][RW][PS], pixel stride PS vectors
template <typename T, int K>
__device__ __forceinline__ void stage_tile(typename Vec<T>::type* tile, const T* in, int in_stride, const DwbArgs& a, const TileId& t, int CGB, int PS) {
    typedef typename Vec<T>::type vec_t;
    constexpr int N = Vec<T>::N, P = K / 2;
    const int RH = a.TH + K - 1, RW = a.TW + K - 1, total = RH * RW * CGB, tid = threadIdx.x;
    for (int base = tid; base < total; base += 256 * 4) {      // 4 independent 16-byte loads in flight per lane
        vec_t v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256;
            const int cgi = idx % CGB, p = idx / CGB;
            const int rx = p % RW, ry = p / RW;
            const int iy = t.y0 - P + ry, ix = t.x0 - P + rx;
            v[u] = (vec_t)(T)0;
            dst[u] = idx < total ? p * PS + cgi : -1;
            if (idx < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v[u] = *reinterpret_cast<const vec_t*>(in + ((size_t)((size_t)t.b * a.H + iy) * a.W + ix) * in_stride + t.c0 + cgi * N);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) tile[dst[u]] = v[u];
    }
}

template <typename T, int K>
__device__ __forceinline__ void stage_weights(typename Vec<T>::type* wl, const T* w, const DwbArgs& a, const TileId& t, int CGB) {
    typedef typename Vec<T>::type vec_t;
    constexpr int N = Vec<T>::N;
    for (int idx = threadIdx.x; idx < K * K * CGB; idx += 256) {
        const int cgi = idx % CGB, kk = idx / CGB;
        wl[idx] = *reinterpret_cast<const vec_t*>(w + (size_t)kk * a.C + t.c0 + cgi * N);      // [k*k][C] (maf_pack_dw)
    }
}

// forward: dst[j] = DW_{k_j}(src[0]) for the NB branches
template <typename T, int K0, int NB>
__global__ __launch_bounds__(256) void dwb_fwd_kernel(const DwbArgs a) {
    constexpr int N = Vec<T>::N, P0 = K0 / 2;
    typedef typename Vec<T>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char dwb_raw[];
    vec_t* tile = reinterpret_cast<vec_t*>(dwb_raw);
    const TileId t = decode_tile(a);
    const int CGB = t.cbe / N, CG = a.CB / N, PS = CG + 2;
    const int RW = a.TW + K0 - 1;
    vec_t* wl = tile + (a.TH + K0 - 1) * RW * PS;              // [NB][K0 * K0 * CG]
    stage_tile<T, K0>(tile, static_cast<const T*>(a.src[0]), a.src_stride[0], a, t, CGB, PS);
    db_static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        stage_weights<T, branch_k<K0, j>()>(wl + j * K0 * K0 * CG, static_cast<const T*>(a.w[j]), a, t, CGB);
    });
    // BatchNorm statistics of the branches (a.stats): per workgroup [NB][2][CB] sums in LDS behind the weights, one global atomic per (branch, channel,
    // statistic) and workgroup at the end — what bn_stats_kernel would compute in a pass of its own over every branch's output
    float* lsum = reinterpret_cast<float*>(wl + NB * K0 * K0 * CG);
    const bool st = a.stats_R > 0;
    if (st)
        for (int i = threadIdx.x; i < NB * 2 * a.CB; i += 256) lsum[i] = 0.f;
    __syncthreads();
    const int NSX = a.TW / R, items = a.TH * NSX * CGB;
    // lanes of a wave that hold the same channel group sit CGB apart: with a power-of-two CGB their statistics are summed with xor shuffles first
    // (one LDS atomic per channel, statistic and wave instead of one per lane — same-address LDS atomics of 32 lanes serialise: measured +1.4 ms per step)
    const bool wave_reduce = CGB < 64 && (CGB & (CGB - 1)) == 0;
    for (int it0 = 0; it0 < items; it0 += 256) {                      // uniform trip count: every lane takes part in the shuffles
        const int it = min(it0 + (int)threadIdx.x, items - 1);
        const int cgi = it % CGB, u = it / CGB, s = u % NSX, ry = u / NSX;
        const int oy = t.y0 + ry, ox0 = t.x0 + s * R;
        const bool live = it0 + (int)threadIdx.x < items && oy < a.H && ox0 < a.W;
        db_static_for<NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value, K = branch_k<K0, j>(), off = P0 - K / 2;
            float acc[R][N];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int q = 0; q < N; ++q) acc[r][q] = 0.f;
            if (live) strip_mac<T, K>(acc, tile + ((ry + off) * RW + s * R + off) * PS + cgi, RW, PS, wl + j * K0 * K0 * CG, CGB, cgi);
            T* out = static_cast<T*>(a.dst[j]) + t.c0 + cgi * N + ((size_t)((size_t)t.b * a.H + oy) * a.W + ox0) * a.dst_stride[j];
            float s0[N], s1[N];
#pragma unroll
            for (int q = 0; q < N; ++q) s0[q] = s1[q] = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (live && ox0 + r < a.W) {
                    vec_t o;
#pragma unroll
                    for (int q = 0; q < N; ++q) {
                        o[q] = (T)acc[r][q];
                        const float f = (float)o[q];                   // the statistics of the STORED (rounded) tensor, as a pass over it would see them
                        s0[q] += f; s1[q] = __builtin_fmaf(f, f, s1[q]);
                    }
                    *reinterpret_cast<vec_t*>(out + (size_t)r * a.dst_stride[j]) = o;
                }
            }
            if (st && a.stats[j]) {                                    // (uniform)
                if (wave_reduce) {
                    for (int o = CGB; o < 64; o <<= 1) {
#pragma unroll
                        for (int q = 0; q < N; ++q) { s0[q] += __shfl_xor(s0[q], o, 64); s1[q] += __shfl_xor(s1[q], o, 64); }
                    }
                }
                if (!wave_reduce ? live : (int)(threadIdx.x & 63) < CGB) {
#pragma unroll
                    for (int q = 0; q < N; ++q) { atomicAdd(&lsum[(j * 2 + 0) * a.CB + cgi * N + q], s0[q]); atomicAdd(&lsum[(j * 2 + 1) * a.CB + cgi * N + q], s1[q]); }
                }
            }
        });
    }
    if (st) {
        __syncthreads();
        const int rep = blockIdx.x % a.stats_R;
        for (int i = threadIdx.x; i < NB * 2 * t.cbe; i += 256) {
            const int j = i / (2 * t.cbe), rem = i - j * 2 * t.cbe, which = rem / t.cbe, c = rem - which * t.cbe;
            if (a.stats[j]) atomicAdd(a.stats[j] + (size_t)rep * 2 * a.C + which * a.C + t.c0 + c, lsum[(j * 2 + which) * a.CB + c]);
        }
    }
}

// data gradient: dst[0] = sum_j DW_{k_j}(src[j]) with the (flipped) filters w[j]; at most IT items per thread (the launcher's tile), sums in registers
template <typename T, int K0, int NB>
__global__ __launch_bounds__(256) void dwb_dgrad_kernel(const DwbArgs a) {
    constexpr int N = Vec<T>::N, IT = 2;
    typedef typename Vec<T>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char dwb_raw[];
    vec_t* tile = reinterpret_cast<vec_t*>(dwb_raw);
    const TileId t = decode_tile(a);
    const int CGB = t.cbe / N, CG = a.CB / N, PS = CG + 2;
    vec_t* wl = tile + (a.TH + K0 - 1) * (a.TW + K0 - 1) * PS;
    const int NSX = a.TW / R, items = a.TH * NSX * CGB;
    float acc[IT][R][N];
    int cgi_[IT], s_[IT], ry_[IT];
    bool ok[IT];
#pragma unroll
    for (int sl = 0; sl < IT; ++sl) {
        const int it = threadIdx.x + sl * 256;
        cgi_[sl] = it % CGB; const int u = it / CGB; s_[sl] = u % NSX; ry_[sl] = u / NSX;
        ok[sl] = it < items && t.y0 + ry_[sl] < a.H && t.x0 + s_[sl] * R < a.W;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < N; ++q) acc[sl][r][q] = 0.f;
    }
    db_static_for<NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value, K = branch_k<K0, j>();
        if constexpr (j > 0) __syncthreads();                  // everybody has finished reading the previous branch's tile
        stage_tile<T, K>(tile, static_cast<const T*>(a.src[j]), a.src_stride[j], a, t, CGB, PS);
        stage_weights<T, K>(wl, static_cast<const T*>(a.w[j]), a, t, CGB);
        __syncthreads();
        const int RW = a.TW + K - 1;
#pragma unroll
        for (int sl = 0; sl < IT; ++sl)
            if (ok[sl]) strip_mac<T, K>(acc[sl], tile + (ry_[sl] * RW + s_[sl] * R) * PS + cgi_[sl], RW, PS, wl, CGB, cgi_[sl]);
    });
#pragma unroll
    for (int sl = 0; sl < IT; ++sl) {
        if (!ok[sl]) continue;
        const int oy = t.y0 + ry_[sl], ox0 = t.x0 + s_[sl] * R;
        T* out = static_cast<T*>(a.dst[0]) + t.c0 + cgi_[sl] * N + ((size_t)((size_t)t.b * a.H + oy) * a.W + ox0) * a.dst_stride[0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (ox0 + r < a.W) {
                vec_t o;
#pragma unroll
                for (int q = 0; q < N; ++q) o[q] = (T)acc[sl][r][q];
                *reinterpret_cast<vec_t*>(out + (size_t)r * a.dst_stride[0]) = o;
            }
        }
    }
}

constexpr size_t kMaxLdsB = 96 * 1024;

size_t lds_bytes(int TH, int TW, int CB, int N, int K0, int NB) {
    return ((size_t)(TH + K0 - 1) * (TW + K0 - 1) * (CB / N + 2) + (size_t)NB * K0 * K0 * (CB / N)) * 16 + (size_t)NB * 2 * CB * sizeof(float);   // tile | weights | statistics
}

// least staged-pixels-per-output-pixel that fits the LDS (the cost model of dwconv.hip + the COVERAGE of the map: 16 x 32 tiles cover a 40 x 40 map
// with 48 x 64 pixels, 1.9x — the first version ignored that and the 40 x 40 data gradients ran at 0.5 TB/s); tile sizes that divide the 20 / 40 /
// 80 / 160-pixel maps are candidates too.  max_items: work items (4-pixel strips x channel groups) a workgroup may hold
void choose_tile(int H, int W, int C, int N, int K0, int NB, int max_items, int& TH, int& TW, int& CB) {
    // measured first (tools/dwb_sweep.py, MI355X, the shapes of MAF-YOLO at 640 px): 32-channel blocks; 10 x 20 tiles on the 20 / 40-pixel maps (no
    // coverage waste), 8 x 32 on the 160-pixel maps, 16 x 16 in between — 190 -> 141 us on 160 x 160 x 72, 77 -> 68 on 80 x 80 x 144, -8 % at 20 x 20;
    // the cost model below only where that tile does not fit
    {
        const int cb = std::min(8 * N, (C + N - 1) / N * N) >= 4 * N ? 4 * N : std::min(8 * N, (C + N - 1) / N * N);
        const int th = (W % 20 == 0 && W <= 40 && H % 10 == 0) ? 10 : W >= 160 ? 8 : 16;
        const int tw = (W % 20 == 0 && W <= 40 && H % 10 == 0) ? 20 : W >= 160 ? 32 : 16;
        if (th <= H && tw <= (W + R - 1) / R * R && lds_bytes(th, tw, cb, N, K0, NB) <= kMaxLdsB && th * (tw / R) * (cb / N) <= max_items) { TH = th; TW = tw; CB = cb; return; }
    }
    const int line = 8 * N;
    double best = 1e30;
    TH = 4; TW = 8; CB = 2 * N;
    const int ths[] = {4, 5, 8, 10, 16, 20, 32, 40}, tws[] = {8, 16, 20, 32, 40};
    for (int cbm = 8; cbm >= 2; cbm >>= 1) {
        const int cb = std::min(cbm * N, (C + N - 1) / N * N);
        for (int th0 : ths) for (int tw0 : tws) {
            const int th = std::min(th0, H), tw = std::min(tw0, (W + R - 1) / R * R);
            const size_t lds = lds_bytes(th, tw, cb, N, K0, NB);
            const int items = th * (tw / R) * (cb / N);
            if (lds > kMaxLdsB || items > max_items) continue;
            const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
            // staged pixels of all tiles (halo clipped at the map's border is still staged as zeros) per pixel of the map
            double cost = (double)ty * tx * (th + K0 - 1) * (tw + K0 - 1) / ((double)H * W);
            if (cb < line && cb < C) cost *= 1.25;
            if (lds > 64 * 1024) cost *= 1.5;
            else if (lds > 40 * 1024) cost *= 1.25;
            else if (lds > 20 * 1024) cost *= 1.1;
            if (items < 256) cost *= 256.0 / items;
            else if (items % 256) cost *= (double)((items + 255) / 256 * 256) / items;      // idle lanes of the last pass
            if (cost < best - 1e-9) { best = cost; TH = th; TW = tw; CB = cb; }
        }
    }
}

template <typename T, int K0, int NB>
int launch_b(DwbArgs& a, bool dgrad, hipStream_t s) {
    constexpr int N = Vec<T>::N;
    choose_tile(a.H, a.W, a.C, N, K0, NB, dgrad ? 512 : (1 << 30), a.TH, a.TW, a.CB);
    if (const char* ov = getenv(dgrad ? "MAF_DWB_TILE_DGRAD" : "MAF_DWB_TILE")) {       // experiments: "rows,cols,channels" for every launch
        int th = 0, tw = 0, cb = 0;
        if (sscanf(ov, "%d,%d,%d", &th, &tw, &cb) == 3 && th > 0 && tw % R == 0 && cb % N == 0 && cb <= 8 * N) {
            th = std::min(th, a.H); tw = std::min(tw, (a.W + R - 1) / R * R); cb = std::min(cb, (a.C + N - 1) / N * N);
            if (lds_bytes(th, tw, cb, N, K0, NB) <= kMaxLdsB && (!dgrad || th * (tw / R) * (cb / N) <= 512)) { a.TH = th; a.TW = tw; a.CB = cb; }
        }
    }
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    a.nwg = a.B * a.tilesY * a.tilesX * a.nCB;
    const size_t lds = lds_bytes(a.TH, a.TW, a.CB, N, K0, NB);
    static bool attr_f = false, attr_d = false;
    bool& attr = dgrad ? attr_d : attr_f;
    if (!attr) {
        const void* fn = dgrad ? reinterpret_cast<const void*>(&dwb_dgrad_kernel<T, K0, NB>) : reinterpret_cast<const void*>(&dwb_fwd_kernel<T, K0, NB>);
        if (int rc = maf_check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsB), "hipFuncSetAttribute(dw_branches)")) return rc;
        attr = true;
    }
    if (dgrad) hipLaunchKernelGGL((dwb_dgrad_kernel<T, K0, NB>), dim3(a.nwg), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dwb_fwd_kernel<T, K0, NB>), dim3(a.nwg), dim3(256), lds, s, a);
    return maf_check_hip(hipGetLastError(), "dw_branches launch");
}

template <typename T>
int launch_k(DwbArgs& a, int k0, int nb, bool dgrad, hipStream_t s) {
    if (k0 == 3 && nb == 3) return launch_b<T, 3, 3>(a, dgrad, s);
    if (k0 == 5 && nb == 3) return launch_b<T, 5, 3>(a, dgrad, s);
    if (k0 == 7 && nb == 3) return launch_b<T, 7, 3>(a, dgrad, s);
    if (k0 == 9 && nb == 4) return launch_b<T, 9, 4>(a, dgrad, s);
    maf_set_error("dw_branches: (k0, branches) must be (3, 3), (5, 3), (7, 3) or (9, 4) — kernel sizes 3, 3, 1 / 5, 3, 1 / 7, 5, 3 / 9, 7, 5, 3");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

static int dw_branches_impl(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                            int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t dgrad, float* const* stats, int32_t replicas,
                            maf_stream_t stream);

extern "C" int maf_dw_branches(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                               int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t dgrad, maf_stream_t stream) {
    return dw_branches_impl(src, src_stride, dst, dst_stride, w, nb, k0, B, H, W, C, dtype, dgrad, nullptr, 0, stream);
}

extern "C" int maf_dw_branches_stats(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                                     int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, float* const* stats, int32_t replicas,
                                     maf_stream_t stream) {
    MAF_REQUIRE(stats && replicas >= 1 && replicas <= 64, "dw_branches_stats: stats = one scratch half per branch (or NULL), 1..64 replicas");
    return dw_branches_impl(src, src_stride, dst, dst_stride, w, nb, k0, B, H, W, C, dtype, 0, stats, replicas, stream);
}

static int dw_branches_impl(const void* const* src, const int32_t* src_stride, void* const* dst, const int32_t* dst_stride, const void* const* w,
                            int32_t nb, int32_t k0, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t dgrad, float* const* stats, int32_t replicas,
                            maf_stream_t stream) {
    MAF_REQUIRE(src && src_stride && dst && dst_stride && w && nb >= 1 && nb <= MAXB, "dw_branches: null argument / 1..4 branches");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "dw_branches: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % N == 0, "dw_branches: C must be a multiple of the 16-byte channel group");
    DwbArgs a = {};
    const int nsrc = dgrad ? nb : 1, ndst = dgrad ? 1 : nb;
    for (int j = 0; j < nb; ++j) {
        MAF_REQUIRE(w[j], "dw_branches: null filter");
        a.w[j] = w[j];
    }
    for (int j = 0; j < nsrc; ++j) {
        MAF_REQUIRE(src[j] && src_stride[j] % N == 0 && src_stride[j] >= C, "dw_branches: null source / stride not a multiple of the channel group");
        a.src[j] = src[j]; a.src_stride[j] = src_stride[j];
    }
    for (int j = 0; j < ndst; ++j) {
        MAF_REQUIRE(dst[j] && dst_stride[j] % N == 0 && dst_stride[j] >= C, "dw_branches: null output / stride not a multiple of the channel group");
        a.dst[j] = dst[j]; a.dst_stride[j] = dst_stride[j];
    }
    a.B = B; a.H = H; a.W = W; a.C = C;
    a.stats_R = stats ? replicas : 0;
    for (int j = 0; j < nb; ++j) a.stats[j] = stats ? stats[j] : nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == MAF_F16 ? launch_k<half_t>(a, k0, nb, dgrad != 0, s) : launch_k<float>(a, k0, nb, dgrad != 0, s);
}
