// The sum of the BatchNorms of a train-form DilatedReparamBlock as ONE apply pass per direction (training only).
//
// Reference: yolov6/layers/common.py:3024-3031 — out = origin_bn(lk_origin(x)) + sum_j dil_bn_j(dil_conv_j(x)): NB (3 or 4) BatchNorm2d in training mode, no
// activation, summed (and RepVGGBlock, common.py:224: ReLU(BN(3x3 s2) + BN(1x1 s2)), two branches with a ReLU on the sum: `act`).  As a chain of fused BatchNorm calls (csrc/bn_act.hip with `residual`, rounds 2-3) every branch cost an apply pass that read its z_j AND
// the running sum and wrote the running sum again (3 NB - 1 tensor passes forward), and its own statistics + apply pass over (z_j, d out) backward (5 NB):
//   forward   out = sum_j (z_j * sc_j + sh_j): NB reads, ONE write                                   (NB + 1 passes; the statistics come from the depth-wise
//             kernel's epilogue, csrc/dw_branches.hip, or from maf_bn_stats for the 1 x 1 branch)
//   backward  the upstream gradient g = d out is the SAME for every branch (no activation in between): one statistics pass accumulates sum g (shared) and
//             sum g * xhat_j for all branches (1 + NB reads), one apply pass writes every dz_j = gamma_j rstd_j (g - mean(g) - xhat_j mean(g xhat_j))
//             (1 + NB reads, NB writes)                                                                (2 + 3 NB passes instead of 5 NB)
// dgamma_j = sum g xhat_j, dbeta_j = sum g (written or added: a gradient-exchange bucket slice).  fp16 / fp32 NHWC views, fp32 arithmetic, 16-byte accesses.
#include "maf_common.h"

namespace {

constexpr int MAXB = 4;
constexpr int kMaxR = 16;
constexpr int kU = 2;                                                // pixels per lane in flight (x up to 5 tensors)

struct BsArgs {
    const void* z[MAXB]; int zs[MAXB];
    const void* dy; int dys;
    void* out; int outs;
    void* dz[MAXB]; int dzs[MAXB];
    int nb, M, C, R, act;                                                // act: MAF_ACT_NONE, or MAF_ACT_RELU on the sum (RepVGGBlock, common.py:224)
    const float* gamma[MAXB]; const float* beta[MAXB];
    float* mean[MAXB]; float* rstd[MAXB];
    float* part[MAXB]; float* part_clear[MAXB]; int clear_n;         // forward: every branch's own statistics scratch (this call's half / the half to clear)
    float eps, momentum; float* rmean[MAXB]; float* rvar[MAXB]; long long* counter[MAXB];
    float* bpart; float* bpart_clear; int bclear_n;                  // backward: [R][1 + nb][C] (sum g | sum g xhat_j)
    float* dgamma[MAXB]; float* dbeta[MAXB]; int acc_affine;
    float* opart; int oR;                                              // forward, optional: the scratch half of the BatchNorm that READS `out` next — [oR][2][C] += {sum out, sum out^2}
};

template <typename T> struct Vec;
template <> struct Vec<half_t> { typedef half8_t type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef f32x4_t type; static constexpr int N = 4; };

// out = sum_j (z_j * sc_j + sh_j).  Prologue (every workgroup, as bn_apply_kernel): the per-channel constants of every branch from its partial sums
// (replicas added in double, in order), workgroup 0 publishes save_mean / save_rstd / running statistics / num_batches_tracked; the grid clears the
// other half of every branch's scratch.
// STATS: the statistics of the ROUNDED sums it stores, for the training-mode BatchNorm that normalises `out` next (UniRepLKNetBlock.norm behind the DilatedReparamBlock,
// common.py:3083): that BatchNorm's own statistics pass read `out` once more, one launch per block (16 per step of MAF-YOLO-n).  Lanes of one channel group do not line
// up across a wave here (the group count is the layer's, not a power of two): every thread parks its sums in LDS — the constants' area, free by then — and one thread
// per (statistic, channel) adds its column and issues the one global atomic (as bn_stats_kernel's backward form, csrc/bn_act.hip).
template <typename T, int NB, bool STATS = false>
__global__ __launch_bounds__(256) void bn_sum_apply_kernel(const BsArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    extern __shared__ float cst[];                                    // [NB][2][C]: sc, sh
    for (int c = threadIdx.x; c < a.C; c += 256) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double s = 0, q = 0;
            for (int r = 0; r < a.R; ++r) { s += a.part[j][(size_t)r * 2 * a.C + c]; q += a.part[j][(size_t)r * 2 * a.C + a.C + c]; }
            const double mu = s / a.M;
            double var = q / a.M - mu * mu;
            if (var < 0) var = 0;
            const float muf = (float)mu, rsf = (float)(1.0 / sqrt(var + (double)a.eps));
            const float sc = rsf * a.gamma[j][c];
            cst[(j * 2) * a.C + c] = sc; cst[(j * 2 + 1) * a.C + c] = a.beta[j][c] - muf * sc;
            if (blockIdx.x == 0) {
                if (c == 0 && a.counter[j]) *a.counter[j] += 1;
                a.mean[j][c] = muf; a.rstd[j][c] = rsf;
                if (a.rmean[j]) {
                    const double unb = a.M > 1 ? var * a.M / (a.M - 1) : var;
                    a.rmean[j][c] = (float)((1.0 - a.momentum) * a.rmean[j][c] + a.momentum * mu);
                    a.rvar[j][c] = (float)((1.0 - a.momentum) * a.rvar[j][c] + a.momentum * unb);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.clear_n; i += gridDim.x * 256) a.part_clear[j][i] = 0.f;
    __syncthreads();
    const int groups = a.C / N, gpb = groups < 256 ? groups : 256, plan = 256 / gpb;
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    T* op = static_cast<T*>(a.out);
    float s0[STATS ? N : 1], s1[STATS ? N : 1];                      // (STATS: groups <= 256, one pass of the loop below)
    if constexpr (STATS) {
#pragma unroll
        for (int q = 0; q < N; ++q) s0[q] = s1[q] = 0.f;
    }
    for (int g0 = 0; g0 < groups; g0 += gpb) {
        const int gi = g0 + threadIdx.x % gpb, pl = threadIdx.x / gpb;
        if (gi >= groups || pl >= plan) continue;
        float sc[NB][N], sh[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            sh[q] = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) { sc[j][q] = cst[(j * 2) * a.C + gi * N + q]; sh[q] += cst[(j * 2 + 1) * a.C + gi * N + q]; }
        }
        const bool relu = a.act == MAF_ACT_RELU;
        auto one = [&](const V (&zv)[NB]) {
            V ov;
#pragma unroll
            for (int q = 0; q < N; ++q) {
                float u = sh[q];
#pragma unroll
                for (int j = 0; j < NB; ++j) u = __builtin_fmaf((float)zv[j][q], sc[j][q], u);
                ov[q] = (T)(relu ? fmaxf(u, 0.f) : u);
                if constexpr (STATS) { const float f = (float)ov[q]; s0[q] += f; s1[q] = __builtin_fmaf(f, f, s1[q]); }
            }
            return ov;
        };
        int m = m0 + pl;
        for (; m + (kU - 1) * plan < m1; m += kU * plan) {
            V zv[kU][NB];
#pragma unroll
            for (int u = 0; u < kU; ++u)
#pragma unroll
                for (int j = 0; j < NB; ++j) zv[u][j] = *reinterpret_cast<const V*>(static_cast<const T*>(a.z[j]) + (size_t)(m + u * plan) * a.zs[j] + gi * N);
#pragma unroll
            for (int u = 0; u < kU; ++u) *reinterpret_cast<V*>(op + (size_t)(m + u * plan) * a.outs + gi * N) = one(zv[u]);
        }
        for (; m < m1; m += plan) {
            V zv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) zv[j] = *reinterpret_cast<const V*>(static_cast<const T*>(a.z[j]) + (size_t)m * a.zs[j] + gi * N);
            *reinterpret_cast<V*>(op + (size_t)m * a.outs + gi * N) = one(zv);
        }
    }
    if constexpr (STATS) {
        __syncthreads();                                              // every thread has read its constants
        float* ld = cst;                                              // [256][2 N] (the launch sized the LDS for it)
#pragma unroll
        for (int q = 0; q < N; ++q) { ld[threadIdx.x * 2 * N + q] = s0[q]; ld[threadIdx.x * 2 * N + N + q] = s1[q]; }      // (idle threads: zeros)
        __syncthreads();
        float* dstp = a.opart + (size_t)(blockIdx.x % a.oR) * 2 * a.C;
        for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
            const int which = i / a.C, c = i - which * a.C, gq = c / N, j = c - gq * N;
            float t = 0.f;
            for (int q = 0; q < plan; ++q) t += ld[(q * gpb + gq) * 2 * N + which * N + j];
            atomicAdd(dstp + (size_t)which * a.C + c, t);
        }
    }
}

// bpart += {sum g, sum g xhat_0, ..., sum g xhat_{NB-1}} per channel; geometry of bn_stats_kernel (channel slices of <= 8 groups x pixel chunks)
template <typename T, int NB>
__global__ __launch_bounds__(256) void bn_sum_bwd_stats_kernel(const BsArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N, NS = 1 + NB;
    extern __shared__ float lsum[];                                   // [NS][gs * N]
    const int groups = a.C / N;
    const int gs = (groups + gridDim.y - 1) / gridDim.y;
    const int gbeg = blockIdx.y * gs, gcnt = min(gs, groups - gbeg), cs = gs * N;
    for (int i = threadIdx.x; i < NS * cs; i += 256) lsum[i] = 0.f;
    __syncthreads();
    const int gsl = gs;                                                               // dense lane map; a group count that is not a power of two parks its sums in LDS (below; csrc/bn_act.hip)
    const int plan = 256 / gsl;
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    const bool wave_reduce = gsl < 64 && (gsl & (gsl - 1)) == 0;
    const int gl = threadIdx.x % gsl, pl = threadIdx.x / gsl, gi = gbeg + gl;
    const bool active = gl < gcnt && pl < plan;
    const bool relu = a.act == MAF_ACT_RELU;                         // g = dy * [u > 0], u = sum_j (xhat_j gamma_j + beta_j) recomputed
    float acc[NS][N], mu[NB][N], rs[NB][N], ga[NB][N], bsum[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        bsum[q] = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k][q] = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            mu[j][q] = active ? a.mean[j][gi * N + q] : 0.f; rs[j][q] = active ? a.rstd[j][gi * N + q] : 0.f;
            ga[j][q] = (active && relu) ? a.gamma[j][gi * N + q] : 0.f;
            if (active && relu) bsum[q] += a.beta[j][gi * N + q];
        }
    }
    if (active) {
        const T* dp = static_cast<const T*>(a.dy);
        for (int m = m0 + pl; m < m1; m += kU * plan) {
            V gv[kU], zv[kU][NB];
            bool ok[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                ok[u] = m + u * plan < m1;
                const size_t mm = ok[u] ? (size_t)(m + u * plan) : (size_t)m;
                gv[u] = *reinterpret_cast<const V*>(dp + mm * a.dys + gi * N);
#pragma unroll
                for (int j = 0; j < NB; ++j) zv[u][j] = *reinterpret_cast<const V*>(static_cast<const T*>(a.z[j]) + mm * a.zs[j] + gi * N);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    float xh[NB], uu = bsum[q];
#pragma unroll
                    for (int j = 0; j < NB; ++j) { xh[j] = ((float)zv[u][j][q] - mu[j][q]) * rs[j][q]; uu = __builtin_fmaf(xh[j], ga[j][q], uu); }
                    const float g = (relu && !(uu > 0.f)) ? 0.f : (float)gv[u][q];
                    acc[0][q] += g;
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[1 + j][q] = __builtin_fmaf(g, xh[j], acc[1 + j][q]);
                }
            }
        }
    }
    if (!wave_reduce) {
        __syncthreads();
        float* ld = lsum;                                             // [256][NS N]
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < N; ++q) ld[threadIdx.x * NS * N + k * N + q] = active ? acc[k][q] : 0.f;
        __syncthreads();
        float* dstp = a.bpart + (size_t)(blockIdx.x % a.R) * NS * a.C + gbeg * N;
        for (int i = threadIdx.x; i < NS * gcnt * N; i += 256) {
            const int k = i / (gcnt * N), c = i - k * gcnt * N, gq = c / N, q = c - gq * N;
            float t = 0.f;
            for (int pq = 0; pq < plan; ++pq) t += ld[(pq * gsl + gq) * NS * N + k * N + q];
            atomicAdd(dstp + (size_t)k * a.C + c, t);
        }
        return;
    }
    if (wave_reduce) {
        for (int off = gsl; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int q = 0; q < N; ++q) acc[k][q] += __shfl_xor(acc[k][q], off, 64);
        }
    }
    if (active && (!wave_reduce || (threadIdx.x & 63) < gsl)) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int q = 0; q < N; ++q) atomicAdd(&lsum[k * cs + gl * N + q], acc[k][q]);
    }
    __syncthreads();
    float* dst = a.bpart + (size_t)(blockIdx.x % a.R) * NS * a.C + gbeg * N;
    for (int i = threadIdx.x; i < NS * gcnt * N; i += 256) {
        const int k = i / (gcnt * N), c = i - k * gcnt * N;
        atomicAdd(dst + (size_t)k * a.C + c, lsum[k * cs + c]);
    }
}

// dz_j = gamma_j rstd_j (g - sum_g / M - xhat_j sum_gx_j / M); workgroup 0 writes (or adds) dgamma_j = sum_gx_j, dbeta_j = sum_g
template <typename T, int NB>
__global__ __launch_bounds__(256) void bn_sum_bwd_apply_kernel(const BsArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N, NS = 1 + NB;
    extern __shared__ float cst[];                                    // [2 + 5 NB][C]: k0 | per branch k1, sc, mu, rs | sum of betas | per branch gamma
    const float invM = 1.f / (float)a.M;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double s[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = 0;
        for (int r = 0; r < a.R; ++r)
#pragma unroll
            for (int k = 0; k < NS; ++k) s[k] += a.bpart[((size_t)r * NS + k) * a.C + c];
        cst[c] = (float)s[0] * invM;
        float bs = 0.f;
        if (a.act == MAF_ACT_RELU)
#pragma unroll
            for (int j = 0; j < NB; ++j) bs += a.beta[j][c];
        cst[(1 + 4 * NB) * a.C + c] = bs;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float rsf = a.rstd[j][c];
            cst[(1 + 4 * j) * a.C + c] = (float)s[1 + j] * invM;
            cst[(2 + 4 * j) * a.C + c] = rsf * a.gamma[j][c];
            cst[(3 + 4 * j) * a.C + c] = a.mean[j][c];
            cst[(4 + 4 * j) * a.C + c] = rsf;
            cst[(2 + 4 * NB + j) * a.C + c] = a.gamma[j][c];
            if (blockIdx.x == 0) {                                    // one writer per channel
                if (a.dbeta[j]) a.dbeta[j][c] = (a.acc_affine ? a.dbeta[j][c] : 0.f) + (float)s[0];
                if (a.dgamma[j]) a.dgamma[j][c] = (a.acc_affine ? a.dgamma[j][c] : 0.f) + (float)s[1 + j];
            }
        }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.bclear_n; i += gridDim.x * 256) a.bpart_clear[i] = 0.f;
    __syncthreads();
    const int groups = a.C / N, gpb = groups < 256 ? groups : 256, plan = 256 / gpb;
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    const T* dp = static_cast<const T*>(a.dy);
    for (int g0 = 0; g0 < groups; g0 += gpb) {
        const int gi = g0 + threadIdx.x % gpb, pl = threadIdx.x / gpb;
        if (gi >= groups || pl >= plan) continue;
        const bool relu = a.act == MAF_ACT_RELU;
        float k0[N], k1[NB][N], sc[NB][N], mu[NB][N], rs[NB][N], ga[NB][N], bsum[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int c = gi * N + q;
            k0[q] = cst[c]; bsum[q] = cst[(1 + 4 * NB) * a.C + c];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                k1[j][q] = cst[(1 + 4 * j) * a.C + c]; sc[j][q] = cst[(2 + 4 * j) * a.C + c]; mu[j][q] = cst[(3 + 4 * j) * a.C + c]; rs[j][q] = cst[(4 + 4 * j) * a.C + c];
                ga[j][q] = cst[(2 + 4 * NB + j) * a.C + c];
            }
        }
        for (int m = m0 + pl; m < m1; m += plan) {
            const V gv = *reinterpret_cast<const V*>(dp + (size_t)m * a.dys + gi * N);
            V zv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) zv[j] = *reinterpret_cast<const V*>(static_cast<const T*>(a.z[j]) + (size_t)m * a.zs[j] + gi * N);
            float xh[NB][N], g[N];
#pragma unroll
            for (int q = 0; q < N; ++q) {
                float uu = bsum[q];
#pragma unroll
                for (int j = 0; j < NB; ++j) { xh[j][q] = ((float)zv[j][q] - mu[j][q]) * rs[j][q]; uu = __builtin_fmaf(xh[j][q], ga[j][q], uu); }
                g[q] = (relu && !(uu > 0.f)) ? 0.f : (float)gv[q];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                V ov;
#pragma unroll
                for (int q = 0; q < N; ++q) ov[q] = (T)(sc[j][q] * (g[q] - k0[q] - xh[j][q] * k1[j][q]));
                *reinterpret_cast<V*>(static_cast<T*>(a.dz[j]) + (size_t)m * a.dzs[j] + gi * N) = ov;
            }
        }
    }
}

// grids: the rules of csrc/bn_act.hip (bn_grid / stats_grid)
int apply_grid(int M, int C, int dtype) {
    const int groups = C / (dtype == MAF_F16 ? 8 : 4), gpb = groups < 256 ? groups : 256, plan = 256 / gpb;
    const int ppl_small = (long long)M * C <= 7000000 ? 8 : 16;
    long long g = ((long long)M + (long long)plan * 16 - 1) / ((long long)plan * 16);
    for (int ppl = 8; g < 1024 && ppl >= ppl_small; ppl >>= 1) g = ((long long)M + (long long)plan * ppl - 1) / ((long long)plan * ppl);
    return (int)(g < 1 ? 1 : g > 8192 ? 8192 : g);
}

dim3 stats_grid(int M, int C, int dtype, int NS, size_t* lds) {
    const int N = dtype == MAF_F16 ? 8 : 4, groups = C / N;
    const int nslice = (groups + 7) / 8, gs = (groups + nslice - 1) / nslice, plan = 256 / gs;
    long long gx = ((long long)M + plan * 32 - 1) / (plan * 32);
    const int ppl_small = (long long)M * C <= 7000000 ? 8 : 16;
    for (int ppl = 16; gx * nslice < 1024 && ppl >= ppl_small; ppl >>= 1) gx = ((long long)M + plan * ppl - 1) / (plan * ppl);
    if (gx > 4096) gx = 4096;
    *lds = (size_t)NS * gs * N * sizeof(float);
    if (gs & (gs - 1)) *lds = (size_t)256 * NS * N * sizeof(float);         // the parking area of a slice whose group count is not a power of two
    return dim3((unsigned)(gx < 1 ? 1 : gx), (unsigned)nslice);
}

int replicas(int C, int R) {
    const int want = 1024 / C > 0 ? 1024 / C : 1;
    int r = R < want ? R : want;
    return r > kMaxR ? kMaxR : r;
}

int check(int nb, int M, int C, int dtype, int R) {
    MAF_REQUIRE(nb >= 2 && nb <= MAXB, "bn_sum: 2..4 branches");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "bn_sum: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(M > 0 && C > 0 && C % N == 0 && C <= 4096, "bn_sum: C must be a multiple of the 16-byte channel group (<= 4096)");
    MAF_REQUIRE(R >= 1 && R <= 64, "bn_sum: replicas 1..64");
    return 0;
}

}  // namespace

// Forward: every branch's scratch half `phase[j]` of part[j] ([2][R][2][roundup(C,256)], the layout of maf_bn_forward) holds its {sum z, sum z^2} already
// (maf_dw_branches_stats / maf_bn_stats); one apply pass.
extern "C" int maf_bn_sum_forward(const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                                  const float* const* gamma, const float* const* beta, float eps, float momentum,
                                  float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                                  void* out, int32_t out_stride, float* const* save_mean, float* const* save_rstd,
                                  float* const* part, int32_t R, const int32_t* phase, int32_t act, maf_stream_t stream) {
    return maf_bn_sum_forward_stats(z, z_stride, nb, M, C, dtype, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, out, out_stride, save_mean, save_rstd,
                                    part, R, phase, act, nullptr, 0, 0, stream);
}

// ... and, when next_part is given, half `next_phase` of next_part ([2][next_R][2][roundup(C,256)], the scratch of the BatchNorm that normalises `out` next) += the
// statistics of the values stored: that BatchNorm's call then is its apply pass alone (maf_bn_forward_ex with statistics ready).
extern "C" int maf_bn_sum_forward_stats(const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                                        const float* const* gamma, const float* const* beta, float eps, float momentum,
                                        float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                                        void* out, int32_t out_stride, float* const* save_mean, float* const* save_rstd,
                                        float* const* part, int32_t R, const int32_t* phase, int32_t act,
                                        float* next_part, int32_t next_R, int32_t next_phase, maf_stream_t stream) {
    if (int rc = check(nb, M, C, dtype, R)) return rc;
    MAF_REQUIRE(act == MAF_ACT_NONE || act == MAF_ACT_RELU, "bn_sum: act must be none or relu");
    MAF_REQUIRE(z && z_stride && gamma && beta && out && save_mean && save_rstd && part && phase, "bn_sum_forward: null argument");
    const int N = dtype == MAF_F16 ? 8 : 4;
    BsArgs a = {};
    a.nb = nb; a.M = M; a.C = C; a.R = replicas(C, R); a.eps = eps; a.momentum = momentum; a.act = act;
    a.out = out; a.outs = out_stride;
    MAF_REQUIRE(out_stride % N == 0, "bn_sum_forward: output stride must be a multiple of the channel group");
    const int half = R * 2 * ((C + 255) / 256 * 256);
    a.clear_n = half;
    for (int j = 0; j < nb; ++j) {
        MAF_REQUIRE(z[j] && z_stride[j] % N == 0 && gamma[j] && beta[j] && save_mean[j] && save_rstd[j] && part[j] && (phase[j] == 0 || phase[j] == 1), "bn_sum_forward: null / misaligned branch argument");
        a.z[j] = z[j]; a.zs[j] = z_stride[j]; a.gamma[j] = gamma[j]; a.beta[j] = beta[j]; a.mean[j] = save_mean[j]; a.rstd[j] = save_rstd[j];
        a.part[j] = part[j] + (size_t)phase[j] * half; a.part_clear[j] = part[j] + (size_t)(1 - phase[j]) * half;
        a.rmean[j] = running_mean ? running_mean[j] : nullptr; a.rvar[j] = running_var ? running_var[j] : nullptr;
        a.counter[j] = num_batches_tracked ? reinterpret_cast<long long*>(num_batches_tracked[j]) : nullptr;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ga = apply_grid(M, C, dtype);
    size_t lds = (size_t)nb * 2 * C * sizeof(float);
    if (next_part) {
        MAF_REQUIRE(next_R >= 1 && next_R <= 64 && (next_phase == 0 || next_phase == 1) && C / N <= 256, "bn_sum_forward: next_part = [2][next_R][2][roundup(C,256)], phase 0 / 1, C / 8 <= 256");
        a.oR = replicas(C, next_R);
        a.opart = next_part + (size_t)next_phase * next_R * 2 * ((C + 255) / 256 * 256);
        const size_t park = (size_t)256 * 2 * N * sizeof(float);
        if (lds < park) lds = park;
    }
#define MAF_BS_FWD(TT, NBV) { if (next_part) hipLaunchKernelGGL((bn_sum_apply_kernel<TT, NBV, true>), dim3(ga), dim3(256), lds, s, a); else hipLaunchKernelGGL((bn_sum_apply_kernel<TT, NBV, false>), dim3(ga), dim3(256), lds, s, a); }
    if (dtype == MAF_F16) { if (nb == 2) MAF_BS_FWD(half_t, 2) else if (nb == 3) MAF_BS_FWD(half_t, 3) else MAF_BS_FWD(half_t, 4) }
    else { if (nb == 2) MAF_BS_FWD(float, 2) else if (nb == 3) MAF_BS_FWD(float, 3) else MAF_BS_FWD(float, 4) }
#undef MAF_BS_FWD
    return maf_check_hip(hipGetLastError(), "bn_sum_forward launch");
}

// Backward: dy = gradient of the sum; bpart = [2][R][1 + nb][roundup(C,256)] fp32 scratch, zeroed once by the caller, halves alternate (phase) like maf_bn_backward's.
extern "C" int maf_bn_sum_backward(const void* dy, int32_t dy_stride, const void* const* z, const int32_t* z_stride, int32_t nb, int32_t M, int32_t C, int32_t dtype,
                                   const float* const* gamma, const float* const* beta, const float* const* save_mean, const float* const* save_rstd,
                                   void* const* dz, const int32_t* dz_stride, float* const* dgamma, float* const* dbeta, int32_t accumulate_affine,
                                   float* bpart, int32_t R, int32_t phase, int32_t act, maf_stream_t stream) {
    if (int rc = check(nb, M, C, dtype, R)) return rc;
    MAF_REQUIRE(act == MAF_ACT_NONE || act == MAF_ACT_RELU, "bn_sum: act must be none or relu");
    MAF_REQUIRE(beta, "bn_sum_backward: null argument");
    MAF_REQUIRE(dy && z && z_stride && gamma && save_mean && save_rstd && dz && dz_stride && dgamma && dbeta && bpart && (phase == 0 || phase == 1), "bn_sum_backward: null argument");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(dy_stride % N == 0, "bn_sum_backward: gradient stride must be a multiple of the channel group");
    BsArgs a = {};
    a.nb = nb; a.M = M; a.C = C; a.R = replicas(C, R); a.dy = dy; a.dys = dy_stride; a.acc_affine = accumulate_affine; a.act = act;
    const int half = R * (1 + nb) * ((C + 255) / 256 * 256);
    a.bpart = bpart + (size_t)phase * half; a.bpart_clear = bpart + (size_t)(1 - phase) * half; a.bclear_n = half;
    for (int j = 0; j < nb; ++j) {
        MAF_REQUIRE(z[j] && z_stride[j] % N == 0 && gamma[j] && save_mean[j] && save_rstd[j] && dz[j] && dz_stride[j] % N == 0, "bn_sum_backward: null / misaligned branch argument");
        MAF_REQUIRE(beta[j], "bn_sum_backward: null beta");
        a.z[j] = z[j]; a.zs[j] = z_stride[j]; a.gamma[j] = gamma[j]; a.beta[j] = beta[j]; a.mean[j] = const_cast<float*>(save_mean[j]); a.rstd[j] = const_cast<float*>(save_rstd[j]);
        a.dz[j] = dz[j]; a.dzs[j] = dz_stride[j]; a.dgamma[j] = dgamma[j]; a.dbeta[j] = dbeta[j];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t lds_s;
    const dim3 gs = stats_grid(M, C, dtype, 1 + nb, &lds_s);
    const int ga = apply_grid(M, C, dtype);
    const size_t la = (size_t)(2 + 5 * nb) * C * sizeof(float);
#define MAF_BS_BWD(TT, NBV)                                                                        \
    {                                                                                              \
        static bool attr = false;                                                                  \
        if (!attr && la > 64 * 1024) {                                                             \
            if (int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_sum_bwd_apply_kernel<TT, NBV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(bn_sum)")) return rc; \
            attr = true;                                                                           \
        }                                                                                          \
        hipLaunchKernelGGL((bn_sum_bwd_stats_kernel<TT, NBV>), gs, dim3(256), lds_s, s, a);        \
        hipLaunchKernelGGL((bn_sum_bwd_apply_kernel<TT, NBV>), dim3(ga), dim3(256), la, s, a);     \
    }
    if (dtype == MAF_F16) { if (nb == 2) MAF_BS_BWD(half_t, 2) else if (nb == 3) MAF_BS_BWD(half_t, 3) else MAF_BS_BWD(half_t, 4) }
    else { if (nb == 2) MAF_BS_BWD(float, 2) else if (nb == 3) MAF_BS_BWD(float, 3) else MAF_BS_BWD(float, 4) }
#undef MAF_BS_BWD
    return maf_check_hip(hipGetLastError(), "bn_sum_backward launch");
}
