// BatchNorm2d (training mode: batch statistics + running-statistics update) fused with the activation that follows it, forward
// and backward, on NHWC views.  The train-form graph of the reference keeps every conv, BatchNorm and activation apart
// (Conv.forward = act(bn(conv(x))), yolov6/layers/common.py:46-47; conv_bn :157-163; DilatedReparamBlock :3024-3031;
// UniRepLKNetBlock.norm :3083), so a training step runs ~140 BatchNorms + ~70 SiLUs as 7-8 memory passes each.  Here, TWO launches
// per direction:
//   forward   bn_stats (sum, sum of squares per channel)  ->  bn_apply (every workgroup folds the partial sums into mean / rstd,
//             workgroup 0 also writes save_mean / save_rstd / the running statistics; then normalise + affine + activation in one pass)
//   backward  g = dz * act'(u), u recomputed from x:  bn_bwd_stats (sum g, sum g*xhat)  ->  bn_bwd_apply (workgroup 0 writes
//             dgamma / dbeta; dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)))
// fp16 or fp32 tensors, fp32 arithmetic, 16-byte vector accesses, four independent loads in flight per lane in every streaming
// loop.  Partial sums go to `R` replicated fp32 buffers (atomics on one cache line serialise).  The scratch is TWO halves
// [2][R][2][roundup(C,256)]: a call accumulates into half `phase` and its apply kernel clears half `1 - phase` — the one the
// previous call on the stream used and everybody has finished with — so one buffer per stream, zeroed once, serves every BatchNorm
// of a step without memsets or finalize launches; the caller alternates `phase`.
#include <cstdlib>
#include "maf_common.h"

namespace {

struct BnArgs2 {
    const void* x; const void* dz; void* y;           // y: forward output / backward dx
    int xs, dzs, ys;                                  // pixel strides in elements
    int M, C, act, R;
    const float* gamma; const float* beta;
    float* mean; float* rstd;                         // forward: written by workgroup 0; backward: read
    float* part;                                      // this call's half: [R][2][C]
    float* part_clear; int clear_n;                   // the other half, zeroed by the apply kernel
    float eps, momentum; float* running_mean; float* running_var; long long* counter;
    float* dgamma; float* dbeta;
    const void* res; void* dres; int rs, drs;          // residual added before the activation (forward, and backward when the activation needs u); d residual out
    int acc_affine;                                   // backward: dgamma / dbeta are ADDED to what the buffers hold (a gradient-exchange bucket slice)
    float* det;                                       // deterministic mode (maf_set_deterministic): per-workgroup sums [gridDim.x][2][C], no atomics anywhere
};

template <typename T> struct Vec;
template <> struct Vec<half_t> { typedef half8_t type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef f32x4_t type; static constexpr int N = 4; };

__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == MAF_ACT_SILU) return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));
    if (act == MAF_ACT_RELU) return u > 0.f ? u : 0.f;
    return u;
}
__device__ __forceinline__ float act_grad(float u, int act) {       // d act / d u
    if (act == MAF_ACT_SILU) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u)); return s * (1.f + u * (1.f - s)); }
    if (act == MAF_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    return 1.f;
}

constexpr int kMaxR = 16;                                            // replicas actually used: min(R, kMaxR, 1024 / C)
constexpr int kU = 4;                                                // independent 16-byte loads in flight per lane and tensor

// BWD = false: part += {sum x, sum x^2};  BWD = true: part += {sum g, sum g*xhat}
// workgroup = a SLICE of at most 8 channel groups (blockIdx.y; 128 B of a pixel row) x a chunk of pixels (blockIdx.x); thread = one
// N-channel group x a strided set of the chunk's pixels.  Reduction: across the lanes of a wave that hold the same channel group (xor
// shuffles, when the slice has a power-of-two group count), LDS atomics, then one global atomic per (channel, statistic) into replica
// blockIdx.x % R.  The global atomics are what bounds this kernel when a workgroup sees too few pixels (~40 G atomics/s chip-wide
// measured, i.e. 2C atomics cost as much as streaming ~250 B per channel): slicing the channels keeps >= 512 pixels per workgroup for
// any C at >= 2 workgroups per CU.
// ACT: the activation as a compile-time constant (round 6).  With `a.act` read at run time the SiLU / ReLU / none choice was a scalar compare + branch per ELEMENT in the
// unrolled streaming loops — 130 branches and 96 s_nop per 32 elements of the backward apply pass, in kernels that are bound by their vector work, not by bytes.
template <typename T, bool BWD, bool RES = false, int ACT = MAF_ACT_NONE>
__global__ __launch_bounds__(256) void bn_stats_kernel(const BnArgs2 a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    extern __shared__ float lsum[];                                   // [2][gs * N]
    const int groups = a.C / N;
    const int gs = (groups + gridDim.y - 1) / gridDim.y;             // channel groups per slice
    const int gbeg = blockIdx.y * gs, gcnt = min(gs, groups - gbeg);
    const int cs = gs * N;                                           // channels per slice (LDS row)
    for (int i = threadIdx.x; i < 2 * cs; i += 256) lsum[i] = 0.f;
    __syncthreads();
    // lane map: gsl = gs rounded up to a power of two (gs <= 8: slices of <= 8 groups) group lanes x 256 / gsl pixel lanes, the group lanes past the slice idle.
    // With gs itself (6 for C = 48 / 96 / 144, 5 for 72, 3 for 24 — the widths of the 160 x 160 and 80 x 80 maps) the lanes of a channel group do not line up
    // across a wave, the xor-shuffle reduction below is off and every lane issued 2 N LDS atomics onto gs N addresses, 42-fold same-address: tools/bn_bench.py —
    // the statistics pass of 32 x 160 x 160 x 48 ran at 2.2 TB/s, of x 128 at 3.9.
    // (the backward pass is bound by its SiLU-gradient arithmetic, where idle lanes cost: it keeps the dense lane map and parks its partial sums in LDS instead — below)
    const int gsl = BWD ? gs : (gs <= 1 ? 1 : gs <= 2 ? 2 : gs <= 4 ? 4 : gs <= 8 ? 8 : gs);
    const int plan = 256 / gsl;                                      // pixel lanes
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    const bool wave_reduce = gsl < 64 && (gsl & (gsl - 1)) == 0;     // lane % gsl == channel group for every wave
    const T* xp = static_cast<const T*>(a.x);
    const T* dp = static_cast<const T*>(a.dz);
    const int gl = threadIdx.x % gsl, pl = threadIdx.x / gsl, gi = gbeg + gl;
    const bool active = gl < gcnt && pl < plan;
    float s0[N], s1[N], mu[N], rs[N], ga[N], be[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        s0[j] = s1[j] = 0.f;
        if (BWD && active) { mu[j] = a.mean[gi * N + j]; rs[j] = a.rstd[gi * N + j]; ga[j] = a.gamma[gi * N + j]; be[j] = a.beta[gi * N + j]; }
    }
    const T* rp = static_cast<const T*>(a.res);
    auto accum = [&](const V& xv, const V& dv, const V& rv) {
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const float f = (float)xv[j]; s0[j] += f; s1[j] = __builtin_fmaf(f, f, s1[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float xh = ((float)xv[j] - mu[j]) * rs[j];
                float u = __builtin_fmaf(xh, ga[j], be[j]);
                if (RES) u += (float)rv[j];
                const float gq = (float)dv[j] * act_grad(u, ACT);
                s0[j] += gq; s1[j] = __builtin_fmaf(gq, xh, s1[j]);
            }
        }
    };
    if (active) {
        int m = m0 + pl;
        for (; m + (kU - 1) * plan < m1; m += kU * plan) {
            V xv[kU], dv[kU], rv[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                xv[u] = *reinterpret_cast<const V*>(xp + (size_t)(m + u * plan) * a.xs + gi * N);
                if (BWD) dv[u] = *reinterpret_cast<const V*>(dp + (size_t)(m + u * plan) * a.dzs + gi * N);
                if (BWD && RES) rv[u] = *reinterpret_cast<const V*>(rp + (size_t)(m + u * plan) * a.rs + gi * N);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) accum(xv[u], dv[u], rv[u]);
        }
        for (; m < m1; m += plan) {
            const V xv = *reinterpret_cast<const V*>(xp + (size_t)m * a.xs + gi * N);
            V dv = xv, rv = xv;
            if (BWD) dv = *reinterpret_cast<const V*>(dp + (size_t)m * a.dzs + gi * N);
            if (BWD && RES) rv = *reinterpret_cast<const V*>(rp + (size_t)m * a.rs + gi * N);
            accum(xv, dv, rv);
        }
    }
    if (a.det) {
        // Bit-reproducible statistics (maf_set_deterministic; a test / debugging mode): no atomics — every thread parks its partial sums in LDS, one
        // thread per (channel, statistic) adds the pixel lanes of its channel IN ORDER, the workgroup's sums go to its own slot with plain stores and
        // bn_det_reduce_kernel adds the slots in order.  (The fp32 atomics of the normal path land in another order every run; one ulp in a
        // BatchNorm's statistics is enough to flip a near-tied arg-max of a max-pool further down, i.e. a discrete change of the BACKWARD pass.)
        __syncthreads();
        float* ld = lsum;                                             // [256][2 N] (the launch sized the LDS for it)
#pragma unroll
        for (int j = 0; j < N; ++j) { ld[threadIdx.x * 2 * N + j] = active ? s0[j] : 0.f; ld[threadIdx.x * 2 * N + N + j] = active ? s1[j] : 0.f; }
        __syncthreads();
        for (int i = threadIdx.x; i < gcnt * N; i += 256) {
            const int gq = i / N, j = i - gq * N;
            double t0 = 0, t1 = 0;
            for (int q = 0; q < plan; ++q) { t0 += ld[(q * gsl + gq) * 2 * N + j]; t1 += ld[(q * gsl + gq) * 2 * N + N + j]; }
            float* slot = a.det + (size_t)blockIdx.x * 2 * a.C + gbeg * N + i;
            slot[0] = (float)t0; slot[a.C] = (float)t1;
        }
        return;
    }
    if (!wave_reduce) {
        // group count of the slice not a power of two (backward pass): the lanes of a channel group do not line up across a wave.  Every thread parks its sums
        // in LDS ([256][2 N]: the launch sized it), one thread per (statistic, channel) adds its column — plan values, conflict-free — and issues the one global atomic.
        __syncthreads();
        float* ld = lsum;
#pragma unroll
        for (int j = 0; j < N; ++j) { ld[threadIdx.x * 2 * N + j] = active ? s0[j] : 0.f; ld[threadIdx.x * 2 * N + N + j] = active ? s1[j] : 0.f; }
        __syncthreads();
        float* dstp = a.part + (size_t)(blockIdx.x % a.R) * 2 * a.C + gbeg * N;
        for (int i = threadIdx.x; i < 2 * gcnt * N; i += 256) {
            const int which = i / (gcnt * N), c = i - which * gcnt * N, gq = c / N, j = c - gq * N;
            float t = 0.f;
            for (int q = 0; q < plan; ++q) t += ld[(q * gsl + gq) * 2 * N + which * N + j];
            atomicAdd(dstp + (size_t)which * a.C + c, t);
        }
        return;
    }
    if (wave_reduce) {
        for (int off = gsl; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < N; ++j) { s0[j] += __shfl_xor(s0[j], off, 64); s1[j] += __shfl_xor(s1[j], off, 64); }
        }
    }
    if (active && (!wave_reduce || (threadIdx.x & 63) < gsl)) {
#pragma unroll
        for (int j = 0; j < N; ++j) { atomicAdd(&lsum[gl * N + j], s0[j]); atomicAdd(&lsum[cs + gl * N + j], s1[j]); }
    }
    __syncthreads();
    float* dst = a.part + (size_t)(blockIdx.x % a.R) * 2 * a.C + gbeg * N;
    for (int i = threadIdx.x; i < gcnt * N; i += 256) { atomicAdd(dst + i, lsum[i]); atomicAdd(dst + a.C + i, lsum[cs + i]); }
}

// deterministic mode: part[replica 0] = the per-workgroup sums added in workgroup order (the other replicas stay zero)
__global__ void bn_det_reduce_kernel(const float* det, int nslot, int C, float* part) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * C) return;
    double t = 0;
    for (int b = 0; b < nslot; ++b) t += det[(size_t)b * 2 * C + i];
    part[i] = (float)t;
}

// BWD = false: y = act(xhat*gamma + beta);  BWD = true: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M)
// prologue: per-channel constants from the partial sums into LDS (every workgroup; the sums are added in double, in replica order, so all
// workgroups agree), workgroup 0 publishes the statistics / parameter gradients; the grid clears the scratch half of the previous call.
// thread = one N-channel group (its constants live in registers) x a strided set of pixels of the workgroup's chunk
template <typename T, bool BWD, bool RES = false, int ACT = MAF_ACT_NONE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs2 a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    extern __shared__ float cst[];                                    // forward [2][C]: sc, sh;  backward [6][C]: mu, rs, ga, be, k0, k1
    const float invM = 1.f / (float)a.M;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float v0[kMaxR], v1[kMaxR];                                   // all replica loads in flight before the first add
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            const float* p0 = a.part + (size_t)r * 2 * a.C + c;
            v0[r] = r < a.R ? p0[0] : 0.f; v1[r] = r < a.R ? p0[a.C] : 0.f;
        }
        double s = 0, q = 0;
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) { s += v0[r]; q += v1[r]; }
        if (!BWD) {
            const double mu = s / a.M;
            double var = q / a.M - mu * mu;                          // biased (normalisation)
            if (var < 0) var = 0;
            const float muf = (float)mu, rsf = (float)(1.0 / sqrt(var + (double)a.eps));
            const float sc = rsf * a.gamma[c];
            cst[c] = sc; cst[a.C + c] = a.beta[c] - muf * sc;         // u = x*sc + sh
            if (blockIdx.x == 0) {
                if (c == 0 && a.counter) *a.counter += 1;             // nn.BatchNorm2d.num_batches_tracked
                a.mean[c] = muf; a.rstd[c] = rsf;
                if (a.running_mean) {                                // torch: running = (1 - momentum) * running + momentum * batch (unbiased var)
                    const double unb = a.M > 1 ? var * a.M / (a.M - 1) : var;
                    a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mu);
                    a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
                }
            }
        } else {
            cst[c] = a.mean[c]; cst[a.C + c] = a.rstd[c]; cst[2 * a.C + c] = a.gamma[c]; cst[3 * a.C + c] = a.beta[c];
            cst[4 * a.C + c] = (float)s * invM; cst[5 * a.C + c] = (float)q * invM;
            if (blockIdx.x == 0) {                                    // one writer per channel: a plain read-modify-write when accumulating
                if (a.dbeta) a.dbeta[c] = (a.acc_affine ? a.dbeta[c] : 0.f) + (float)s;
                if (a.dgamma) a.dgamma[c] = (a.acc_affine ? a.dgamma[c] : 0.f) + (float)q;
            }
        }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.clear_n; i += gridDim.x * 256) a.part_clear[i] = 0.f;
    __syncthreads();
    const int groups = a.C / N;
    const int gpb = groups < 256 ? groups : 256;
    const int plan = 256 / gpb;
    // (walking the tensor from its END — the tail is what the 256 MB memory-side cache still holds of the statistics pass — measured no different)
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    const T* xp = static_cast<const T*>(a.x);
    const T* dp = static_cast<const T*>(a.dz);
    T* yp = static_cast<T*>(a.y);
    for (int g0 = 0; g0 < groups; g0 += gpb) {
        const int gi = g0 + threadIdx.x % gpb, pl = threadIdx.x / gpb;
        if (gi >= groups || pl >= plan) continue;
        float sc[N], sh[N], ga[N], be[N], mu[N], rs[N], k0[N], k1[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int c = gi * N + j;
            if (!BWD) { sc[j] = cst[c]; sh[j] = cst[a.C + c]; }
            else {
                mu[j] = cst[c]; rs[j] = cst[a.C + c]; ga[j] = cst[2 * a.C + c]; be[j] = cst[3 * a.C + c]; k0[j] = cst[4 * a.C + c]; k1[j] = cst[5 * a.C + c];
                sc[j] = rs[j] * ga[j];
            }
        }
        const T* rp = static_cast<const T*>(a.res);
        T* drp = static_cast<T*>(a.dres);
        auto one = [&](const V& xv, const V& dv, const V& rv, V& gv) {
            V ov;
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float u = __builtin_fmaf((float)xv[j], sc[j], sh[j]);
                    if (RES) u += (float)rv[j];
                    ov[j] = (T)act_fwd(u, ACT);
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    float u = __builtin_fmaf(xh, ga[j], be[j]);
                    if (RES) u += (float)rv[j];
                    const float gq = (float)dv[j] * act_grad(u, ACT);
                    if (RES) gv[j] = (T)gq;                          // gradient of the residual input
                    ov[j] = (T)(sc[j] * (gq - k0[j] - xh * k1[j]));
                }
            }
            return ov;
        };
        int m = m0 + pl;
        for (; m + (kU - 1) * plan < m1; m += kU * plan) {
            V xv[kU], dv[kU], rv[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                xv[u] = *reinterpret_cast<const V*>(xp + (size_t)(m + u * plan) * a.xs + gi * N);
                if (BWD) dv[u] = *reinterpret_cast<const V*>(dp + (size_t)(m + u * plan) * a.dzs + gi * N);
                if (RES) rv[u] = *reinterpret_cast<const V*>(rp + (size_t)(m + u * plan) * a.rs + gi * N);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                V gv;
                *reinterpret_cast<V*>(yp + (size_t)(m + u * plan) * a.ys + gi * N) = one(xv[u], dv[u], rv[u], gv);
                if (BWD && RES) *reinterpret_cast<V*>(drp + (size_t)(m + u * plan) * a.drs + gi * N) = gv;
            }
        }
        for (; m < m1; m += plan) {
            const V xv = *reinterpret_cast<const V*>(xp + (size_t)m * a.xs + gi * N);
            V dv = xv, rv = xv, gv;
            if (BWD) dv = *reinterpret_cast<const V*>(dp + (size_t)m * a.dzs + gi * N);
            if (RES) rv = *reinterpret_cast<const V*>(rp + (size_t)m * a.rs + gi * N);
            *reinterpret_cast<V*>(yp + (size_t)m * a.ys + gi * N) = one(xv, dv, rv, gv);
            if (BWD && RES) *reinterpret_cast<V*>(drp + (size_t)m * a.drs + gi * N) = gv;
        }
    }
}

int check_common(const void* x, int32_t xs, int32_t M, int32_t C, int32_t dtype, int32_t R, int32_t phase, const float* part) {
    MAF_REQUIRE(x && M > 0 && C > 0, "bn: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "bn: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && xs % N == 0, "bn: C and strides must be multiples of the 16-byte channel group");
    MAF_REQUIRE(C <= 4096, "bn: C too large");
    MAF_REQUIRE(part && R >= 1 && R <= 64 && (phase == 0 || phase == 1), "bn: part = [2][R][2][roundup(C,256)] fp32 (zeroed once), replicas 1..64, phase 0 / 1");
    return 0;
}

// ~16 pixels per lane and pass; at most `cap` workgroups.  (Measured: 4 pixels per lane — four times the workgroups — is SLOWER on
// every shape, 17.6 -> 75 us on 32x20x20x768: the per-workgroup prologue / atomics, not the streaming loop, is the fixed cost.)
int bn_grid(int M, int C, int dtype, int cap) {
    const int groups = C / (dtype == MAF_F16 ? 8 : 4), gpb = groups < 256 ? groups : 256, plan = 256 / gpb;
    const int ppl_small = (long long)M * C <= 7000000 ? 8 : 16;            // see stats_grid
    long long g = ((long long)M + (long long)plan * 16 - 1) / ((long long)plan * 16);
    for (int ppl = 8; g < 1024 && ppl >= ppl_small; ppl >>= 1) g = ((long long)M + (long long)plan * ppl - 1) / ((long long)plan * ppl);
    return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

// statistics pass: channel slices of <= 8 groups (blockIdx.y), 32 pixels per lane (16 when that leaves fewer than ~4 workgroups per CU)
dim3 stats_grid(int M, int C, int dtype, size_t* lds) {
    const int groups = C / (dtype == MAF_F16 ? 8 : 4), N = dtype == MAF_F16 ? 8 : 4;
    const int nslice = (groups + 7) / 8, gs = (groups + nslice - 1) / nslice, plan = 256 / (gs <= 1 ? 1 : gs <= 2 ? 2 : gs <= 4 ? 4 : 8);      // pixel lanes of bn_stats_kernel's lane map
    long long gx = ((long long)M + plan * 32 - 1) / (plan * 32);
    // tensors of <= 7 M elements (the 20 x 20 maps, narrow 40 x 40 ones) are latency-bound — 100 workgroups, each lane walking 16 pixels of
    // SiLU-gradient arithmetic at one wave per SIMD: 8 pixels per lane there (32x20x20x256: 16.5 / 30.2 -> 13.3 / 21.0 us forward /
    // backward); on anything bigger more, shorter workgroups cost more than they hide (80x80x96: 47 / 64 -> 51 / 74 us)
    const int ppl_small = (long long)M * C <= 7000000 ? 8 : 16;
    for (int ppl = 16; gx * nslice < 1024 && ppl >= ppl_small; ppl >>= 1) gx = ((long long)M + plan * ppl - 1) / (plan * ppl);
    if (gx > 4096) gx = 4096;
    *lds = (size_t)2 * gs * N * sizeof(float);
    if (gs & (gs - 1)) *lds = (size_t)256 * 2 * N * sizeof(float);          // the backward statistics kernel parks its per-thread sums (dense lane map)
    return dim3((unsigned)(gx < 1 ? 1 : gx), (unsigned)nslice);
}

bool g_det = false;
constexpr int kMaxDev = 16;
float* g_det_bufs[kMaxDev] = {};                                     // one slot buffer per device (a process may drive several)
size_t g_det_caps[kMaxDev] = {};

// deterministic mode: the slot buffer (grow-only, owned by the library) and the launch geometry of the statistics kernel
int det_prepare(BnArgs2& a, const dim3& gs, int C, int dtype, size_t* lds, hipStream_t s) {
    a.det = nullptr;
    if (!g_det) return 0;
    int dev = 0;
    if (int rc = maf_check_hip(hipGetDevice(&dev), "bn deterministic: hipGetDevice")) return rc;
    MAF_REQUIRE(dev >= 0 && dev < kMaxDev, "bn deterministic: device index out of range");
    float*& g_det_buf = g_det_bufs[dev];
    size_t& g_det_cap = g_det_caps[dev];
    const size_t need = (size_t)gs.x * 2 * C * sizeof(float);
    if (need > g_det_cap) {
        if (int rc = maf_check_hip(hipDeviceSynchronize(), "bn deterministic: hipDeviceSynchronize")) return rc;
        if (g_det_buf) (void)hipFree(g_det_buf);
        g_det_buf = nullptr; g_det_cap = 0;
        if (int rc = maf_check_hip(hipMalloc(reinterpret_cast<void**>(&g_det_buf), need * 2), "bn deterministic: hipMalloc")) return rc;
        g_det_cap = need * 2;
    }
    a.det = g_det_buf;
    *lds = (size_t)256 * 2 * (dtype == MAF_F16 ? 8 : 4) * sizeof(float);
    return 0;
}

void det_reduce(const BnArgs2& a, const dim3& gs, int C, hipStream_t s) {
    if (a.det) hipLaunchKernelGGL(bn_det_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, a.det, (int)gs.x, C, a.part);
}

void set_halves(BnArgs2& a, float* part, int C, int R, int phase) {
    const int half = R * 2 * ((C + 255) / 256 * 256);
    // wide layers have few workgroups per address and every workgroup of the apply kernel reads all replicas: use fewer of them
    const int want = 1024 / C > 0 ? 1024 / C : 1;
    a.R = R < want ? R : want;
    if (a.R > kMaxR) a.R = kMaxR;
    a.part = part + (size_t)phase * half;
    a.part_clear = part + (size_t)(1 - phase) * half;
    a.clear_n = half;
}

}  // namespace

extern "C" int maf_set_deterministic(int32_t on) {
    g_det = on != 0;
    return 0;
}

// replicas of the partial sums a call with `R` requested replicas really uses for C channels (set_halves): who fills `part` from another kernel
// (csrc/dw_branches.hip's statistics epilogue) must spread over exactly these
extern "C" int32_t maf_bn_replicas(int32_t C, int32_t R) {
    const int want = 1024 / C > 0 ? 1024 / C : 1;
    int r = R < want ? R : want;
    return r > kMaxR ? kMaxR : r;
}

extern "C" int maf_bn_forward(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t act, void* y,
                              int32_t y_stride, float* save_mean, float* save_rstd, float* part, int32_t R, int32_t phase, const void* residual, int32_t res_stride,
                              maf_stream_t stream) {
    return maf_bn_forward_ex(x, x_stride, M, C, dtype, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, act, y, y_stride, save_mean, save_rstd,
                             part, R, phase, residual, res_stride, 0, stream);
}

extern "C" int maf_bn_forward_ex(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t act, void* y,
                                 int32_t y_stride, float* save_mean, float* save_rstd, float* part, int32_t R, int32_t phase, const void* residual, int32_t res_stride,
                                 int32_t stats_ready, maf_stream_t stream) {
    if (int rc = check_common(x, x_stride, M, C, dtype, R, phase, part)) return rc;
    MAF_REQUIRE(gamma && beta && y && save_mean && save_rstd, "bn_forward: null pointer");
    MAF_REQUIRE(!residual || res_stride % (dtype == MAF_F16 ? 8 : 4) == 0, "bn_forward: residual stride must be a multiple of the 16-byte channel group");
    MAF_REQUIRE(act == MAF_ACT_NONE || act == MAF_ACT_SILU || act == MAF_ACT_RELU, "bn_forward: act must be none / relu / silu");
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnArgs2 a = {};
    a.x = x; a.y = y; a.xs = x_stride; a.ys = y_stride; a.M = M; a.C = C; a.act = act; a.R = R;
    a.mean = save_mean; a.rstd = save_rstd; a.gamma = gamma; a.beta = beta;
    a.eps = eps; a.momentum = momentum; a.running_mean = running_mean; a.running_var = running_var; a.counter = reinterpret_cast<long long*>(num_batches_tracked);
    set_halves(a, part, C, R, phase);
    size_t lds_s;
    const dim3 gs = stats_grid(M, C, dtype, &lds_s);
    const int ga = bn_grid(M, C, dtype, 8192);
    if (!stats_ready) {                                      // else: the producing kernel has accumulated {sum x, sum x^2} into half `phase` already
        if (int rc = det_prepare(a, gs, C, dtype, &lds_s, s)) return rc;
        if (dtype == MAF_F16) hipLaunchKernelGGL((bn_stats_kernel<half_t, false>), gs, dim3(256), lds_s, s, a);
        else hipLaunchKernelGGL((bn_stats_kernel<float, false>), gs, dim3(256), lds_s, s, a);
        det_reduce(a, gs, C, s);
    }
    a.res = residual; a.rs = res_stride;
    const size_t la = (size_t)2 * C * sizeof(float);
#define MAF_BN_FWD(T_, RES_, ACT_) hipLaunchKernelGGL((bn_apply_kernel<T_, false, RES_, ACT_>), dim3(ga), dim3(256), la, s, a)
#define MAF_BN_FWD_A(T_, RES_) do { if (act == MAF_ACT_SILU) MAF_BN_FWD(T_, RES_, MAF_ACT_SILU); else if (act == MAF_ACT_RELU) MAF_BN_FWD(T_, RES_, MAF_ACT_RELU); else MAF_BN_FWD(T_, RES_, MAF_ACT_NONE); } while (0)
    if (residual) {
        if (dtype == MAF_F16) MAF_BN_FWD_A(half_t, true); else MAF_BN_FWD_A(float, true);
    } else {
        if (dtype == MAF_F16) MAF_BN_FWD_A(half_t, false); else MAF_BN_FWD_A(float, false);
    }
#undef MAF_BN_FWD_A
#undef MAF_BN_FWD
    return maf_check_hip(hipGetLastError(), "bn_forward launch");
}

// The statistics pass of maf_bn_forward alone: half `phase` of `part` += {sum x, sum x^2} (for maf_bn_sum_forward's branches whose producer has no
// statistics epilogue).  Nothing is cleared here: the apply pass that reads the half clears the other one.
extern "C" int maf_bn_stats(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, float* part, int32_t R, int32_t phase, maf_stream_t stream) {
    if (int rc = check_common(x, x_stride, M, C, dtype, R, phase, part)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnArgs2 a = {};
    a.x = x; a.xs = x_stride; a.M = M; a.C = C; a.R = R;
    set_halves(a, part, C, R, phase);
    size_t lds_s;
    const dim3 gs = stats_grid(M, C, dtype, &lds_s);
    if (int rc = det_prepare(a, gs, C, dtype, &lds_s, s)) return rc;
    if (dtype == MAF_F16) hipLaunchKernelGGL((bn_stats_kernel<half_t, false>), gs, dim3(256), lds_s, s, a);
    else hipLaunchKernelGGL((bn_stats_kernel<float, false>), gs, dim3(256), lds_s, s, a);
    det_reduce(a, gs, C, s);
    return maf_check_hip(hipGetLastError(), "bn_stats launch");
}

extern "C" int maf_bn_backward(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, int32_t M, int32_t C, int32_t dtype,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int32_t act,
                               void* dx, int32_t dx_stride, float* dgamma, float* dbeta, float* part, int32_t R, int32_t phase,
                               const void* residual, int32_t res_stride, void* dres, int32_t dres_stride, maf_stream_t stream) {
    return maf_bn_backward_acc(x, x_stride, dz, dz_stride, M, C, dtype, gamma, beta, save_mean, save_rstd, act, dx, dx_stride, dgamma, dbeta, part, R, phase,
                               residual, res_stride, dres, dres_stride, 0, stream);
}

extern "C" int maf_bn_backward_acc(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, int32_t M, int32_t C, int32_t dtype,
                                   const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int32_t act,
                                   void* dx, int32_t dx_stride, float* dgamma, float* dbeta, float* part, int32_t R, int32_t phase,
                                   const void* residual, int32_t res_stride, void* dres, int32_t dres_stride, int32_t accumulate_affine, maf_stream_t stream) {
    if (int rc = check_common(x, x_stride, M, C, dtype, R, phase, part)) return rc;
    MAF_REQUIRE(dz && gamma && beta && save_mean && save_rstd && dx, "bn_backward: null pointer");
    MAF_REQUIRE((residual == nullptr) == (dres == nullptr), "bn_backward: residual and its gradient buffer go together (an activation-free BatchNorm passes dz through: no residual here)");
    MAF_REQUIRE(!residual || (res_stride % (dtype == MAF_F16 ? 8 : 4) == 0 && dres_stride % (dtype == MAF_F16 ? 8 : 4) == 0), "bn_backward: residual strides must be multiples of the 16-byte channel group");
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnArgs2 a = {};
    a.x = x; a.dz = dz; a.y = dx; a.xs = x_stride; a.dzs = dz_stride; a.ys = dx_stride; a.M = M; a.C = C; a.act = act; a.R = R;
    a.mean = const_cast<float*>(save_mean); a.rstd = const_cast<float*>(save_rstd); a.gamma = gamma; a.beta = beta;
    a.dgamma = dgamma; a.dbeta = dbeta; a.acc_affine = accumulate_affine;
    set_halves(a, part, C, R, phase);
    size_t lds_s;
    const dim3 gs = stats_grid(M, C, dtype, &lds_s);
    const int ga = bn_grid(M, C, dtype, 8192);
    a.res = residual; a.rs = res_stride; a.dres = dres; a.drs = dres_stride;
    const size_t la = (size_t)6 * C * sizeof(float);
    if (int rc = det_prepare(a, gs, C, dtype, &lds_s, s)) return rc;
#define MAF_BN_BWD(T_, RES_, ACT_) do { hipLaunchKernelGGL((bn_stats_kernel<T_, true, RES_, ACT_>), gs, dim3(256), lds_s, s, a); det_reduce(a, gs, C, s); \
                                        hipLaunchKernelGGL((bn_apply_kernel<T_, true, RES_, ACT_>), dim3(ga), dim3(256), la, s, a); } while (0)
#define MAF_BN_BWD_A(T_, RES_) do { if (act == MAF_ACT_SILU) MAF_BN_BWD(T_, RES_, MAF_ACT_SILU); else if (act == MAF_ACT_RELU) MAF_BN_BWD(T_, RES_, MAF_ACT_RELU); else MAF_BN_BWD(T_, RES_, MAF_ACT_NONE); } while (0)
    if (residual) {
        if (dtype == MAF_F16) MAF_BN_BWD_A(half_t, true); else MAF_BN_BWD_A(float, true);
    } else {
        if (dtype == MAF_F16) MAF_BN_BWD_A(half_t, false); else MAF_BN_BWD_A(float, false);
    }
#undef MAF_BN_BWD_A
#undef MAF_BN_BWD
    return maf_check_hip(hipGetLastError(), "bn_backward launch");
}
