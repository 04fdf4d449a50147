// MaxPool2d(k, stride, padding) for the training graph — SPPF.m, three chained 5 x 5 stride-1 pools on the 20 x 20 map
// (yolov6/layers/common.py:114-129), and MP, the 2 x 2 stride-2 pool of MPRep (:667-673, :776-792) — forward with the argmax kept as one
// byte per element, backward as a GATHER.
//
// The framework's backward scatters with atomics over the overlapping windows: 271 us per pool on 32 x 192 x 20 x 20 (0.8 ms of a
// 31 ms step for three tiny maps).  Here an input element looks at the k*k outputs whose window contains it and takes the gradient
// of those that chose it: k*k (8-byte index + 16-byte gradient) vector loads out of L2 per 8 channels, no atomics.
// Tie rule = the framework's (aten max_pool2d_with_indices): the window is scanned row by row over its in-image part and an element
// replaces the running maximum only if it is greater (or NaN) — the FIRST maximum wins.
#include "maf_common.h"

namespace {

struct MpArgs {
    const void* x; void* y; unsigned char* idx; const void* dy; void* dx;
    int B, H, W, C, xs, ys, k;
    int Ho, Wo, stride, pad;                            // x / dx are [B,H,W,C], y / dy / idx [B,Ho,Wo,C]
};

template <typename T> struct MpVec;
template <> struct MpVec<half_t> { typedef half8_t type; static constexpr int N = 8; };
template <> struct MpVec<float> { typedef f32x4_t type; static constexpr int N = 4; };

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const MpArgs a) {
    typedef typename MpVec<T>::type V;
    constexpr int N = MpVec<T>::N;
    const int CG = a.C / N;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)a.B * a.Ho * a.Wo * CG) return;
    const int cg = (int)(tid % CG);
    long long t = tid / CG;
    const int w = (int)(t % a.Wo); t /= a.Wo;
    const int h = (int)(t % a.Ho);
    const int b = (int)(t / a.Ho);
    const T* xp = static_cast<const T*>(a.x) + cg * N;
    float m[N];
    int am[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { m[j] = -INFINITY; am[j] = -1; }
    for (int i = 0; i < a.k; ++i) {
        const int ih = h * a.stride - a.pad + i;
        if ((unsigned)ih >= (unsigned)a.H) continue;
        for (int jx = 0; jx < a.k; ++jx) {
            const int iw = w * a.stride - a.pad + jx;
            if ((unsigned)iw >= (unsigned)a.W) continue;
            const V v = *reinterpret_cast<const V*>(xp + ((size_t)((size_t)b * a.H + ih) * a.W + iw) * a.xs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float f = (float)v[j];
                if (f > m[j] || f != f || am[j] < 0) { m[j] = f; am[j] = i * a.k + jx; }      // first in-image element, then strictly greater (or NaN)
            }
        }
    }
    const size_t pix = ((size_t)b * a.Ho + h) * a.Wo + w;
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (T)m[j];
    *reinterpret_cast<V*>(static_cast<T*>(a.y) + pix * a.ys + cg * N) = o;
    unsigned char* ip = a.idx + pix * a.C + cg * N;
#pragma unroll
    for (int j = 0; j < N; ++j) ip[j] = (unsigned char)am[j];
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const MpArgs a) {
    typedef typename MpVec<T>::type V;
    constexpr int N = MpVec<T>::N;
    const int CG = a.C / N;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)a.B * a.H * a.W * CG) return;
    const int cg = (int)(tid % CG);
    long long t = tid / CG;
    const int w = (int)(t % a.W); t /= a.W;
    const int h = (int)(t % a.H);
    const int b = (int)(t / a.H);
    float g[N];
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.f;
    for (int i = 0; i < a.k; ++i) {                         // output (oh, ow) sees this input as element (i, jx) of its window
        const int nh = h + a.pad - i;                       // = oh * stride
        if (nh < 0 || nh % a.stride) continue;
        const int oh = nh / a.stride;
        if (oh >= a.Ho) continue;
        for (int jx = 0; jx < a.k; ++jx) {
            const int nw = w + a.pad - jx;
            if (nw < 0 || nw % a.stride) continue;
            const int ow = nw / a.stride;
            if (ow >= a.Wo) continue;
            const size_t pix = ((size_t)b * a.Ho + oh) * a.Wo + ow;
            const unsigned char* ip = a.idx + pix * a.C + cg * N;
            const V dv = *reinterpret_cast<const V*>(static_cast<const T*>(a.dy) + pix * a.ys + cg * N);
            const int want = i * a.k + jx;
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (ip[j] == want) g[j] += (float)dv[j];
        }
    }
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (T)g[j];
    *reinterpret_cast<V*>(static_cast<T*>(a.dx) + (((size_t)b * a.H + h) * a.W + w) * a.xs + cg * N) = o;
}

int mp_check(int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, int32_t dtype, int32_t s0, int32_t s1) {
    MAF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad shape");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "maxpool: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(k >= 2 && k <= 15 && stride >= 1 && stride <= k && pad >= 0 && 2 * pad <= k, "maxpool: kernel 2..15, stride 1..k, padding <= k/2 (the argmax is one byte)");
    MAF_REQUIRE(H + 2 * pad >= k && W + 2 * pad >= k, "maxpool: window larger than the padded input");
    MAF_REQUIRE(C % N == 0 && s0 % N == 0 && s1 % N == 0, "maxpool: C and pixel strides must be multiples of the 16-byte channel group");
    MAF_REQUIRE((long long)B * H * W * (C / N) < (1ll << 31) * 256, "maxpool: too many elements");
    return 0;
}

void mp_fill(MpArgs& a, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad) {
    a.B = B; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;          // floor mode, like nn.MaxPool2d's default
}

// nn.Upsample(scale_factor=2, mode="nearest") of the neck (configs/yaml/MAF-YOLO-*.yaml: two per model) on NHWC views with pixel strides: forward one thread = one
// 16-byte channel chunk of an INPUT pixel stored to its four output pixels (the source may be a slot of a concat buffer, and so may the result: the framework's
// kernel first copies a strided source and cannot store into a slice); backward the sum of the four.
template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void up2_fwd_kernel(const T* __restrict__ x, int xs, T* __restrict__ y, int ys, int H, int W, int CG, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i % CG);
    long long p = i / CG;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const long long b = p / H;
    const V v = *reinterpret_cast<const V*>(x + ((b * H + h) * W + w) * xs + cg * N);
    T* o = y + ((b * 2 * H + 2 * h) * (2ll * W) + 2 * w) * ys + cg * N;
    *reinterpret_cast<V*>(o) = v;
    *reinterpret_cast<V*>(o + ys) = v;
    *reinterpret_cast<V*>(o + 2ll * W * ys) = v;
    *reinterpret_cast<V*>(o + 2ll * W * ys + ys) = v;
}

template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void up2_bwd_kernel(const T* __restrict__ dy, int dys, T* __restrict__ dx, int dxs, int H, int W, int CG, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i % CG);
    long long p = i / CG;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const long long b = p / H;
    const T* q = dy + ((b * 2 * H + 2 * h) * (2ll * W) + 2 * w) * dys + cg * N;
    const V a = *reinterpret_cast<const V*>(q), c = *reinterpret_cast<const V*>(q + dys), d = *reinterpret_cast<const V*>(q + 2ll * W * dys), e = *reinterpret_cast<const V*>(q + 2ll * W * dys + dys);
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (T)(((float)a[j] + (float)c[j]) + ((float)d[j] + (float)e[j]));
    *reinterpret_cast<V*>(dx + ((b * H + h) * W + w) * dxs + cg * N) = o;
}

int up2_check(const void* a, const void* b, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t s0, int32_t s1) {
    MAF_REQUIRE(a && b && B > 0 && H > 0 && W > 0 && C > 0, "upsample2x: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "upsample2x: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && s0 % N == 0 && s1 % N == 0, "upsample2x: C and pixel strides must be multiples of the 16-byte channel group");
    return 0;
}

}  // namespace

// y [B,2H,2W,C] = nearest-neighbour x2 of x [B,H,W,C]
extern "C" int maf_upsample2x_forward(const void* x, int32_t x_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* y, int32_t y_stride, maf_stream_t stream) {
    if (int rc = up2_check(x, y, B, H, W, C, dtype, x_stride, y_stride)) return rc;
    const int N = dtype == MAF_F16 ? 8 : 4;
    const long long total = (long long)B * H * W * (C / N);
    const dim3 g((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL((up2_fwd_kernel<half_t, half8_t, 8>), g, dim3(256), 0, s, static_cast<const half_t*>(x), x_stride, static_cast<half_t*>(y), y_stride, H, W, C / N, total);
    else hipLaunchKernelGGL((up2_fwd_kernel<float, f32x4_t, 4>), g, dim3(256), 0, s, static_cast<const float*>(x), x_stride, static_cast<float*>(y), y_stride, H, W, C / N, total);
    return maf_check_hip(hipGetLastError(), "upsample2x forward launch");
}

// dx [B,H,W,C] = sum of the four dy [B,2H,2W,C] pixels of every input pixel
extern "C" int maf_upsample2x_backward(const void* dy, int32_t dy_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* dx, int32_t dx_stride, maf_stream_t stream) {
    if (int rc = up2_check(dy, dx, B, H, W, C, dtype, dy_stride, dx_stride)) return rc;
    const int N = dtype == MAF_F16 ? 8 : 4;
    const long long total = (long long)B * H * W * (C / N);
    const dim3 g((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL((up2_bwd_kernel<half_t, half8_t, 8>), g, dim3(256), 0, s, static_cast<const half_t*>(dy), dy_stride, static_cast<half_t*>(dx), dx_stride, H, W, C / N, total);
    else hipLaunchKernelGGL((up2_bwd_kernel<float, f32x4_t, 4>), g, dim3(256), 0, s, static_cast<const float*>(dy), dy_stride, static_cast<float*>(dx), dx_stride, H, W, C / N, total);
    return maf_check_hip(hipGetLastError(), "upsample2x backward launch");
}

// y [B,Ho,Wo,C] = maxpool(x [B,H,W,C]), Ho = floor((H + 2 pad - k) / stride) + 1; idx [B,Ho,Wo,C] uint8 = window element (row * k + column) chosen
extern "C" int maf_maxpool_forward(const void* x, int32_t x_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                                   int32_t dtype, void* y, int32_t y_stride, uint8_t* idx, maf_stream_t stream) {
    if (int rc = mp_check(B, H, W, C, k, stride, pad, dtype, x_stride, y_stride)) return rc;
    MAF_REQUIRE(x && y && idx, "maxpool_forward: null pointer");
    MpArgs a = {};
    mp_fill(a, B, H, W, C, k, stride, pad);
    a.x = x; a.y = y; a.idx = idx; a.xs = x_stride; a.ys = y_stride;
    const long long total = (long long)B * a.Ho * a.Wo * (C / (dtype == MAF_F16 ? 8 : 4));
    const dim3 g((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<half_t>, g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, g, dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "maxpool forward launch");
}

// dx [B,H,W,C] (pixel stride dx_stride) from dy [B,Ho,Wo,C] (pixel stride dy_stride) and the forward's idx
extern "C" int maf_maxpool_backward(const void* dy, int32_t dy_stride, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                                    int32_t stride, int32_t pad, int32_t dtype, void* dx, int32_t dx_stride, maf_stream_t stream) {
    if (int rc = mp_check(B, H, W, C, k, stride, pad, dtype, dx_stride, dy_stride)) return rc;
    MAF_REQUIRE(dy && dx && idx, "maxpool_backward: null pointer");
    MpArgs a = {};
    mp_fill(a, B, H, W, C, k, stride, pad);
    a.dy = dy; a.dx = dx; a.idx = const_cast<uint8_t*>(idx); a.xs = dx_stride; a.ys = dy_stride;
    const long long total = (long long)B * H * W * (C / (dtype == MAF_F16 ? 8 : 4));
    const dim3 g((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<half_t>, g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, g, dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "maxpool backward launch");
}
