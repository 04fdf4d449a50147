// BatchNorm2d (training mode: batch statistics + running-statistics update) fused with the activation that follows it, forward
// and backward, on NHWC views.  The train-form graph of the reference keeps every conv, BatchNorm and activation apart
// (Conv.forward = act(bn(conv(x))), yolov6/layers/common.py:46-47; conv_bn :157-163; DilatedReparamBlock :3024-3031;
// UniRepLKNetBlock.norm :3083), so a training step runs ~140 BatchNorms + ~70 SiLUs as 7-8 memory passes each.  Here:
//   forward   bn_stats (sum, sum of squares per channel)  ->  bn_finalize (mean, rstd, running stats)  ->  bn_apply (normalise +
//             affine + activation in one pass)
//   backward  g = dz * act'(u), u recomputed from x:  bn_bwd_stats (sum g, sum g*xhat)  ->  bn_bwd_finalize (dgamma, dbeta)  ->
//             bn_bwd_apply (dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)))
// fp16 or fp32 tensors, fp32 arithmetic, 16-byte vector accesses; partial sums go to `R` replicated fp32 buffers (atomics on one
// cache line serialise) that the finalize kernels add up in double and clear again: `part` must be zero on entry and IS zero on
// exit, so one scratch buffer per stream serves every BatchNorm of a step without a memset.
#include "maf_common.h"

namespace {

struct BnArgs2 {
    const void* x; const void* dz; void* y;           // y: forward output / backward dx
    int xs, dzs, ys;                                  // pixel strides in elements
    int M, C, act, R;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    float* part;                                      // [R][2][C]
    const float* sums;                                // [2][C] (backward apply)
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

// BWD = false: part += {sum x, sum x^2};  BWD = true: part += {sum g, sum g*xhat}
// thread = one N-channel group x a strided set of pixels; per-workgroup reduction through LDS atomics, then one global atomic per
// (channel, statistic) into replica blockIdx.x % R
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const BnArgs2 a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    extern __shared__ float lsum[];                                   // [2][C]
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) lsum[i] = 0.f;
    __syncthreads();
    const int groups = a.C / N;
    const int gpb = groups < 256 ? groups : 256;                     // channel groups handled per pass
    const int plan = 256 / gpb;                                      // pixel lanes
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    for (int g0 = 0; g0 < groups; g0 += gpb) {
        const int gi = g0 + threadIdx.x % gpb, pl = threadIdx.x / gpb;
        if (gi >= groups || pl >= plan) continue;
        float s0[N], s1[N], mu[N], rs[N], ga[N], be[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            s0[j] = s1[j] = 0.f;
            if (BWD) { mu[j] = a.mean[gi * N + j]; rs[j] = a.rstd[gi * N + j]; ga[j] = a.gamma[gi * N + j]; be[j] = a.beta[gi * N + j]; }
        }
        for (int m = m0 + pl; m < m1; m += plan) {
            const V xv = *reinterpret_cast<const V*>(static_cast<const T*>(a.x) + (size_t)m * a.xs + gi * N);
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < N; ++j) { const float f = (float)xv[j]; s0[j] += f; s1[j] = __builtin_fmaf(f, f, s1[j]); }
            } else {
                const V dv = *reinterpret_cast<const V*>(static_cast<const T*>(a.dz) + (size_t)m * a.dzs + gi * N);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    const float gq = (float)dv[j] * act_grad(__builtin_fmaf(xh, ga[j], be[j]), a.act);
                    s0[j] += gq; s1[j] = __builtin_fmaf(gq, xh, s1[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) { atomicAdd(&lsum[gi * N + j], s0[j]); atomicAdd(&lsum[a.C + gi * N + j], s1[j]); }
    }
    __syncthreads();
    float* dst = a.part + (size_t)(blockIdx.x % a.R) * 2 * a.C;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) atomicAdd(dst + i, lsum[i]);
}

// forward finalize: one thread per channel
// (the partial sums are cleared as they are read: the caller's scratch buffer stays zeroed for the next BatchNorm on the stream)
__global__ void bn_finalize_kernel(float* part, int R, int M, int C, float eps, float momentum, float* running_mean, float* running_var,
                                   float* mean, float* rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int r = 0; r < R; ++r) {
        float* p0 = part + (size_t)r * 2 * C + c;
        s += p0[0]; q += p0[C];
        p0[0] = 0.f; p0[C] = 0.f;
    }
    const double mu = s / M;
    double var = q / M - mu * mu;                                    // biased (normalisation)
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {                                              // torch: running = (1 - momentum) * running + momentum * batch (unbiased var)
        const double unb = M > 1 ? var * M / (M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

// backward finalize: sums[0][c] = sum g = dbeta, sums[1][c] = sum g*xhat = dgamma
__global__ void bn_bwd_finalize_kernel(float* part, int R, int C, float* sums, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int r = 0; r < R; ++r) {
        float* p0 = part + (size_t)r * 2 * C + c;
        s += p0[0]; q += p0[C];
        p0[0] = 0.f; p0[C] = 0.f;
    }
    sums[c] = (float)s; sums[C + c] = (float)q;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
}

// BWD = false: y = act(xhat*gamma + beta);  BWD = true: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M)
// thread = one N-channel group (its per-channel constants live in registers) x a strided set of pixels of the workgroup's chunk
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs2 a) {
    typedef typename Vec<T>::type V;
    constexpr int N = Vec<T>::N;
    const int groups = a.C / N;
    const int gpb = groups < 256 ? groups : 256;
    const int plan = 256 / gpb;
    const int chunk = (a.M + gridDim.x - 1) / gridDim.x;
    const int m0 = blockIdx.x * chunk, m1 = min(a.M, m0 + chunk);
    const float invM = 1.f / (float)a.M;
    for (int g0 = 0; g0 < groups; g0 += gpb) {
        const int gi = g0 + threadIdx.x % gpb, pl = threadIdx.x / gpb;
        if (gi >= groups || pl >= plan) continue;
        float sc[N], sh[N], ga[N], be[N], mu[N], rs[N], k0[N], k1[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int c = gi * N + j;
            mu[j] = a.mean[c]; rs[j] = a.rstd[c]; ga[j] = a.gamma[c]; be[j] = a.beta[c];
            sc[j] = rs[j] * ga[j]; sh[j] = be[j] - mu[j] * sc[j];                 // u = x*sc + sh
            if (BWD) { k0[j] = a.sums[c] * invM; k1[j] = a.sums[a.C + c] * invM; }
        }
        for (int m = m0 + pl; m < m1; m += plan) {
            const V xv = *reinterpret_cast<const V*>(static_cast<const T*>(a.x) + (size_t)m * a.xs + gi * N);
            V ov;
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < N; ++j) ov[j] = (T)act_fwd(__builtin_fmaf((float)xv[j], sc[j], sh[j]), a.act);
            } else {
                const V dv = *reinterpret_cast<const V*>(static_cast<const T*>(a.dz) + (size_t)m * a.dzs + gi * N);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * rs[j];
                    const float gq = (float)dv[j] * act_grad(__builtin_fmaf(xh, ga[j], be[j]), a.act);
                    ov[j] = (T)(sc[j] * (gq - k0[j] - xh * k1[j]));
                }
            }
            *reinterpret_cast<V*>(static_cast<T*>(a.y) + (size_t)m * a.ys + gi * N) = ov;
        }
    }
}

int check_common(const void* x, int32_t xs, int32_t M, int32_t C, int32_t dtype) {
    MAF_REQUIRE(x && M > 0 && C > 0, "bn: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "bn: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && xs % N == 0, "bn: C and strides must be multiples of the 16-byte channel group");
    MAF_REQUIRE(C <= 8192, "bn: C too large");
    return 0;
}

int stats_grid(int M) { const int g = (M + 255) / 256; return g < 1024 ? (g > 0 ? g : 1) : 1024; }
int apply_grid(int M, int C, int dtype) {                            // ~16 pixels per lane and pass, at most 8192 workgroups
    const int groups = C / (dtype == MAF_F16 ? 8 : 4), gpb = groups < 256 ? groups : 256, plan = 256 / gpb;
    const long long g = ((long long)M + (long long)plan * 16 - 1) / ((long long)plan * 16);
    return (int)(g < 1 ? 1 : g > 8192 ? 8192 : g);
}

}  // namespace

extern "C" int maf_bn_forward(const void* x, int32_t x_stride, int32_t M, int32_t C, int32_t dtype, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, int32_t act, void* y, int32_t y_stride,
                              float* save_mean, float* save_rstd, float* part, int32_t R, maf_stream_t stream) {
    if (int rc = check_common(x, x_stride, M, C, dtype)) return rc;
    MAF_REQUIRE(gamma && beta && y && save_mean && save_rstd && part && R >= 1 && R <= 64, "bn_forward: null pointer / replicas 1..64 (part = [R][2][C] zeroed)");
    MAF_REQUIRE(act == MAF_ACT_NONE || act == MAF_ACT_SILU || act == MAF_ACT_RELU, "bn_forward: act must be none / relu / silu");
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnArgs2 a = {};
    a.x = x; a.y = y; a.xs = x_stride; a.ys = y_stride; a.M = M; a.C = C; a.act = act; a.R = R;
    a.mean = save_mean; a.rstd = save_rstd; a.gamma = gamma; a.beta = beta; a.part = part;
    const size_t lds = (size_t)2 * C * sizeof(float);
    const int gs = stats_grid(M);
    if (dtype == MAF_F16) hipLaunchKernelGGL((bn_stats_kernel<half_t, false>), dim3(gs), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((bn_stats_kernel<float, false>), dim3(gs), dim3(256), lds, s, a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, s, part, R, M, C, eps, momentum, running_mean, running_var, save_mean, save_rstd);
    const int ga = apply_grid(M, C, dtype);
    if (dtype == MAF_F16) hipLaunchKernelGGL((bn_apply_kernel<half_t, false>), dim3(ga), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_apply_kernel<float, false>), dim3(ga), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "bn_forward launch");
}

extern "C" int maf_bn_backward(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, int32_t M, int32_t C, int32_t dtype,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int32_t act,
                               void* dx, int32_t dx_stride, float* dgamma, float* dbeta, float* part, int32_t R, float* sums, maf_stream_t stream) {
    if (int rc = check_common(x, x_stride, M, C, dtype)) return rc;
    MAF_REQUIRE(dz && gamma && beta && save_mean && save_rstd && dx && part && sums && R >= 1 && R <= 64, "bn_backward: null pointer / replicas 1..64 (part = [R][2][C] zeroed, sums = [2][C])");
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnArgs2 a = {};
    a.x = x; a.dz = dz; a.y = dx; a.xs = x_stride; a.dzs = dz_stride; a.ys = dx_stride; a.M = M; a.C = C; a.act = act; a.R = R;
    a.mean = save_mean; a.rstd = save_rstd; a.gamma = gamma; a.beta = beta; a.part = part; a.sums = sums;
    const size_t lds = (size_t)2 * C * sizeof(float);
    const int gs = stats_grid(M);
    if (dtype == MAF_F16) hipLaunchKernelGGL((bn_stats_kernel<half_t, true>), dim3(gs), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((bn_stats_kernel<float, true>), dim3(gs), dim3(256), lds, s, a);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, s, part, R, C, sums, dgamma, dbeta);
    const int ga = apply_grid(M, C, dtype);
    if (dtype == MAF_F16) hipLaunchKernelGGL((bn_apply_kernel<half_t, true>), dim3(ga), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bn_apply_kernel<float, true>), dim3(ga), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "bn_backward launch");
}
