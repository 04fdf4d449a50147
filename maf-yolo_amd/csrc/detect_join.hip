// Detect_yaml's TRAIN branch (yolov6/models/yolo.py:333-354) and the head's class sigmoid (yolov6/layers/common.py:1332) as one launch per direction.
//
// The reference does, per level l:  cls_l = sigmoid(cls_pred(...)) [B,nc,h,w];  cls_l.flatten(2).permute(0,2,1) -> [B,h*w,nc];  the same for reg_l [B,4*(reg_max+1),h,w];
// then torch.cat over the levels -> cls [B,A,nc], reg [B,A,4*(reg_max+1)].  The head tensors here are NHWC in memory, i.e. flatten + permute is the identity on
// the bytes of a level and the whole branch is a strided copy of three maps into one [B,A,C] tensor with the sigmoid applied on the way (11 framework launches
// before: 3 sigmoids, 6 copies of torch.cat, their backward twins).  Backward: d logits = d cls * y (1 - y) with y the joined probabilities (what torch's sigmoid
// backward computes from its saved OUTPUT), d reg copied; both scattered back into per-level NHWC gradient maps whose channel count is padded to the 16-byte
// group the conv kernels read (68 -> 72 for reg at fp16): the pad channels are written as zeros here, so no memset and no F.pad follows.
//
// HBM-bound copy: thread = one 4-element channel group (8 B fp16 / 16 B fp32; nc = 80 and 68 are both multiples of 4, 68 is not one of 8) of one anchor, groups of an
// anchor fastest, so a wave covers contiguous 160 B / 136 B runs.  n at batch 32: 80 MB in + 80 MB out per direction.
#include "maf_common.h"

namespace {

constexpr int kMaxLevels = 4;

struct JoinArgs {
    const void* cls[kMaxLevels];
    const void* reg[kMaxLevels];
    int cs[kMaxLevels], rs[kMaxLevels];      // pixel strides of the level maps, in elements
    int hw[kMaxLevels], a0[kMaxLevels];      // pixels per image of a level, first anchor of the level
    int nl, A, gc, gr;                       // levels, anchors per image, 4-element groups per anchor of cls / reg
    void* cls_out;
    void* reg_out;
    long long total;
};

template <typename T>
__device__ __forceinline__ float sigmoid_of(float x);
template <>
__device__ __forceinline__ float sigmoid_of<half_t>(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }   // rounded to fp16 afterwards
template <>
__device__ __forceinline__ float sigmoid_of<float>(float x) { return 1.f / (1.f + expf(-x)); }                      // the fp32 parity path: IEEE divide, libm exponential

template <typename T, typename V>
__global__ __launch_bounds__(256) void detect_join_kernel(const JoinArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int G = a.gc + a.gr;
    const int g = (int)(i % G);
    const long long m = i / G;               // b * A + anchor
    const int an = (int)(m % a.A);
    const long long b = m / a.A;
    // the level of this anchor: constant indices into the kernel arguments + selects (a per-lane index would send the argument arrays through scratch)
    const void* cp = a.cls[0];
    const void* rp = a.reg[0];
    int cs = a.cs[0], rs = a.rs[0], hw = a.hw[0], a0 = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k)
        if (k < a.nl && an >= a.a0[k]) { cp = a.cls[k]; rp = a.reg[k]; cs = a.cs[k]; rs = a.rs[k]; hw = a.hw[k]; a0 = a.a0[k]; }
    const long long p = b * hw + (an - a0);
    if (g < a.gc) {
        const V v = *reinterpret_cast<const V*>(static_cast<const T*>(cp) + p * cs + g * 4);
        V o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (T)sigmoid_of<T>((float)v[j]);
        *reinterpret_cast<V*>(static_cast<T*>(a.cls_out) + m * (a.gc * 4) + g * 4) = o;
    } else {
        const int gg = g - a.gc;
        *reinterpret_cast<V*>(static_cast<T*>(a.reg_out) + m * (a.gr * 4) + gg * 4) = *reinterpret_cast<const V*>(static_cast<const T*>(rp) + p * rs + gg * 4);
    }
}

struct JoinBwdArgs {
    void* dcls[kMaxLevels];
    void* dreg[kMaxLevels];
    int cs[kMaxLevels], rs[kMaxLevels];
    int hw[kMaxLevels], a0[kMaxLevels];
    int nl, A, gc, gr, gcp, grp;             // gcp / grp: groups per pixel of the gradient maps INCLUDING their zero pad
    const void* d_cls;                       // [B,A,nc] or null (no gradient: zeros)
    const void* d_reg;
    const void* y;                           // the joined probabilities
    long long total;
};

template <typename T, typename V>
__global__ __launch_bounds__(256) void detect_join_bwd_kernel(const JoinBwdArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int G = a.gcp + a.grp;
    const int g = (int)(i % G);
    const long long m = i / G;
    const int an = (int)(m % a.A);
    const long long b = m / a.A;
    void* cp = a.dcls[0];
    void* rp = a.dreg[0];
    int cs = a.cs[0], rs = a.rs[0], hw = a.hw[0], a0 = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k)
        if (k < a.nl && an >= a.a0[k]) { cp = a.dcls[k]; rp = a.dreg[k]; cs = a.cs[k]; rs = a.rs[k]; hw = a.hw[k]; a0 = a.a0[k]; }
    const long long p = b * hw + (an - a0);
    V o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)0.f;
    if (g < a.gcp) {
        if (g < a.gc && a.d_cls) {
            const long long off = m * (a.gc * 4) + g * 4;
            const V d = *reinterpret_cast<const V*>(static_cast<const T*>(a.d_cls) + off);
            const V y = *reinterpret_cast<const V*>(static_cast<const T*>(a.y) + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float yy = (float)y[j]; o[j] = (T)((float)d[j] * (1.f - yy) * yy); }     // torch: grad * (1 - y) * y
        }
        *reinterpret_cast<V*>(static_cast<T*>(cp) + p * cs + g * 4) = o;
    } else {
        const int gg = g - a.gcp;
        if (gg < a.gr && a.d_reg) o = *reinterpret_cast<const V*>(static_cast<const T*>(a.d_reg) + m * (a.gr * 4) + gg * 4);
        *reinterpret_cast<V*>(static_cast<T*>(rp) + p * rs + gg * 4) = o;
    }
}

}  // namespace

extern "C" int maf_detect_join(const void* const* cls, const int32_t* cls_stride, const void* const* reg, const int32_t* reg_stride, const int32_t* level_pixels,
                               int32_t n_levels, int32_t B, int32_t nc, int32_t nreg, int32_t dtype, void* cls_out, void* reg_out, maf_stream_t stream) {
    MAF_REQUIRE(cls && cls_stride && reg && reg_stride && level_pixels && cls_out && reg_out, "detect_join: null argument");
    MAF_REQUIRE(n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && nc > 0 && nreg > 0, "detect_join: 1..4 levels, positive sizes");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "detect_join: dtype must be f16/f32");
    MAF_REQUIRE(nc % 4 == 0 && nreg % 4 == 0, "detect_join: channel counts must be multiples of 4");
    JoinArgs a = {};
    int A = 0;
    for (int l = 0; l < n_levels; ++l) {
        MAF_REQUIRE(cls[l] && reg[l] && level_pixels[l] > 0 && cls_stride[l] >= nc && reg_stride[l] >= nreg && cls_stride[l] % 4 == 0 && reg_stride[l] % 4 == 0,
                    "detect_join: level map null / pixel stride smaller than the channel count or not a multiple of 4");
        a.cls[l] = cls[l]; a.reg[l] = reg[l]; a.cs[l] = cls_stride[l]; a.rs[l] = reg_stride[l]; a.hw[l] = level_pixels[l]; a.a0[l] = A;
        A += level_pixels[l];
    }
    a.nl = n_levels; a.A = A; a.gc = nc / 4; a.gr = nreg / 4; a.cls_out = cls_out; a.reg_out = reg_out;
    a.total = (long long)B * A * (a.gc + a.gr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((a.total + 255) / 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((detect_join_kernel<half_t, half4_t>), g, b, 0, s, a);
    else hipLaunchKernelGGL((detect_join_kernel<float, f32x4_t>), g, b, 0, s, a);
    return maf_check_hip(hipGetLastError(), "detect_join launch");
}

extern "C" int maf_detect_join_backward(const void* d_cls, const void* d_reg, const void* cls_out, const int32_t* level_pixels, int32_t n_levels, int32_t B, int32_t nc,
                                        int32_t nreg, int32_t dtype, void* const* dcls, const int32_t* dcls_stride, void* const* dreg, const int32_t* dreg_stride,
                                        int32_t nc_pad, int32_t nreg_pad, maf_stream_t stream) {
    MAF_REQUIRE(level_pixels && dcls && dcls_stride && dreg && dreg_stride && (cls_out || !d_cls), "detect_join_backward: null argument");
    MAF_REQUIRE(n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && nc > 0 && nreg > 0, "detect_join_backward: 1..4 levels, positive sizes");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "detect_join_backward: dtype must be f16/f32");
    MAF_REQUIRE(nc % 4 == 0 && nreg % 4 == 0 && nc_pad % 4 == 0 && nreg_pad % 4 == 0 && nc_pad >= nc && nreg_pad >= nreg, "detect_join_backward: channel counts / pads must be multiples of 4");
    JoinBwdArgs a = {};
    int A = 0;
    for (int l = 0; l < n_levels; ++l) {
        MAF_REQUIRE(dcls[l] && dreg[l] && level_pixels[l] > 0 && dcls_stride[l] >= nc_pad && dreg_stride[l] >= nreg_pad && dcls_stride[l] % 4 == 0 && dreg_stride[l] % 4 == 0,
                    "detect_join_backward: gradient map null / pixel stride smaller than the padded channel count or not a multiple of 4");
        a.dcls[l] = dcls[l]; a.dreg[l] = dreg[l]; a.cs[l] = dcls_stride[l]; a.rs[l] = dreg_stride[l]; a.hw[l] = level_pixels[l]; a.a0[l] = A;
        A += level_pixels[l];
    }
    a.nl = n_levels; a.A = A; a.gc = nc / 4; a.gr = nreg / 4; a.gcp = nc_pad / 4; a.grp = nreg_pad / 4;
    a.d_cls = d_cls; a.d_reg = d_reg; a.y = cls_out;
    a.total = (long long)B * A * (a.gcp + a.grp);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((a.total + 255) / 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((detect_join_bwd_kernel<half_t, half4_t>), g, b, 0, s, a);
    else hipLaunchKernelGGL((detect_join_bwd_kernel<float, f32x4_t>), g, b, 0, s, a);
    return maf_check_hip(hipGetLastError(), "detect_join_backward launch");
}
