// The loss terms of ComputeLoss on the device, forward and gradient, from the assigner's two per-anchor arrays.
// Replaces, for the task-aligned branch of ComputeLoss.__call__ (yolov6/models/loss.py:56-193):
//   loss_decode_kernel   bbox_decode (:190-193: softmax over the 17 bins, expectation, dist2bbox) * stride -> the pixel boxes the
//                        assigner ranks;
//   loss_cls_kernel      VarifocalLoss (:196-206) over all B*A*nc scores: weight = 0.75 p^2 (1 - y) + t y, BCE(p, t) * weight, with
//                        t = norm of the assigned box on its class and 0 elsewhere — the one-hot labels and the [B,A,nc] score
//                        target of the reference are never built;
//   loss_box_kernel      BboxLoss (:209-267) on the foreground anchors: GIoU loss (figure_iou.py, eps 1e-10) and Distribution Focal
//                        Loss, both weighted by the anchor's target score.
//   loss_finish_kernel   adds the per-workgroup sums (deterministic, no atomics) -> out[5] = weighted total, the three weighted items
//                        (iou, dfl, cls: the order ComputeLoss returns them in) and the target-score sum.
// Both term kernels come in a sums-only form (forward) and a form that writes the gradient with respect to the head outputs, scaled
// by upstream * loss weight / target-score sum read from device memory: nothing is ever read back to the host.
#include "maf_common.h"

namespace {

constexpr int kR1 = 17;               // reg_max + 1 bins per box side
constexpr int kTB = 256;

template <typename T> struct Vec4;    // 4 elements: 8 bytes of f16, 16 bytes of f32
template <> struct Vec4<_Float16> { typedef uint2 type; };
template <> struct Vec4<float> { typedef uint4 type; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- decode: 64 anchors per workgroup staged through LDS with 4-element vector loads, one thread per (anchor, side)
template <typename T>
__global__ __launch_bounds__(kTB) void loss_decode_kernel(const T* __restrict__ distri, const float* __restrict__ pts, const float* __restrict__ st,
                                                          int NA, int A, float* __restrict__ out) {
    typedef typename Vec4<T>::type V;
    __shared__ V tile[64 * kR1];                                                 // 64 anchors x 68 logits
    const int tid = threadIdx.x;
    const size_t a0 = (size_t)blockIdx.x * 64;
    const int nv = (int)min((size_t)64, (size_t)NA - a0) * kR1;
    const V* src = reinterpret_cast<const V*>(distri) + a0 * kR1;
    for (int v = tid; v < nv; v += kTB) tile[v] = src[v];
    __syncthreads();
    const int al = tid >> 2, side = tid & 3;
    const size_t an = a0 + al;
    if (an >= (size_t)NA) return;
    const T* z = reinterpret_cast<const T*>(tile) + al * 4 * kR1 + side * kR1;
    float e[kR1], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kR1; ++k) { e[k] = (float)z[k]; m = fmaxf(m, e[k]); }
    float sum = 0.f, acc = 0.f;
#pragma unroll
    for (int k = 0; k < kR1; ++k) { const float p = expf(e[k] - m); sum += p; acc += p * (float)k; }
    const float d = acc / sum;
    const int ai = (int)(an % (size_t)A);
    const float s = st[ai], c = pts[2 * ai + (side & 1)] / s;
    out[an * 4 + side] = (side < 2 ? c - d : c + d) * s;
}

// ---- classification term: 8 scores per thread
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<_Float16>(const _Float16* p, float* f) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const _Float16* h = reinterpret_cast<const _Float16*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<_Float16>(_Float16* p, const float* f) {
    uint4 v; _Float16* h = reinterpret_cast<_Float16*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)f[i];
    *reinterpret_cast<uint4*>(p) = v;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
    reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// log / reciprocal on the transcendental unit (v_log_f32 * ln 2, v_rcp_f32: ~1 ulp each) — the sum over 21.5 M scores is VALU-bound
// with the IEEE-exact library forms, and the tests pin the result to the reference at 5e-5 either way.
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }

template <typename T, int VEC, bool GRAD>
__global__ __launch_bounds__(kTB) void loss_cls_kernel(const T* __restrict__ scores, const float* __restrict__ gts, const int* __restrict__ agt,
                                                       const float* __restrict__ norm, int nvec, int nc, float w_cls, const float* __restrict__ upstream,
                                                       const float* __restrict__ fwd_out, float4* __restrict__ partials, T* __restrict__ grad) {
    __shared__ float red[kTB / 64];
    float loss = 0.f;
    const float sc = GRAD ? upstream[0] * w_cls / fwd_out[4] : 0.f;
    for (int v = blockIdx.x * kTB + threadIdx.x; v < nvec; v += gridDim.x * kTB) {
        const int e0 = v * VEC;
        const int an = e0 / nc;
        const int c0 = e0 - an * nc;
        const int gi = agt[an];
        float p[VEC], g[VEC];
        if (VEC == 8) load8<T>(scores + e0, p); else p[0] = (float)scores[e0];
        // every score as a negative first (target 0, weight 0.75 p^2): loss = -0.75 p^2 log(1-p)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float l1 = fmaxf(fast_log(1.f - p[i]), -100.f);                // F.binary_cross_entropy clamps its logs at -100
            const float w = 0.75f * p[i] * p[i];
            loss -= l1 * w;
            if (GRAD) g[i] = (w * p[i] * __builtin_amdgcn_rcpf(fmaxf(p[i] * (1.f - p[i]), 1e-12f)) - 1.5f * p[i] * l1) * sc;
        }
        if (gi >= 0) {                                                            // foreground anchor (about 1 %): redo the score of its class
            const int k = (int)gts[(size_t)gi * 5] - c0;
            if (k >= 0 && k < VEC) {
                float pk = p[0];
#pragma unroll
                for (int i = 1; i < VEC; ++i) pk = i == k ? p[i] : pk;
                const float t = norm[an];
                const float l1 = fmaxf(logf(1.f - pk), -100.f), l0 = fmaxf(logf(pk), -100.f);
                loss += 0.75f * pk * pk * fmaxf(fast_log(1.f - pk), -100.f);     // take the negative's term back
                loss += -(t * l0 + (1.f - t) * l1) * t;                          // BCE(p, t) * weight t
                if (GRAD) {
                    const float gk = (pk - t) / fmaxf(pk * (1.f - pk), 1e-12f) * t * sc;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] = i == k ? gk : g[i];
                }
            }
        }
        if (GRAD) { if (VEC == 8) store8<T>(grad + e0, g); else grad[e0] = (T)g[0]; }
    }
    if (partials == nullptr) return;
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = make_float4(red[0] + red[1] + red[2] + red[3], 0.f, 0.f, 0.f);
}

// adds the per-workgroup rows: out = (weighted total, w_iou * iou, w_dfl * dfl, w_cls * cls, target-score sum)
__global__ __launch_bounds__(kTB) void loss_finish_kernel(const float4* __restrict__ partials, int rows, float w_cls, float w_iou, float w_dfl,
                                                          float* __restrict__ out) {
    __shared__ float red[kTB / 64][4];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = threadIdx.x; r < rows; r += kTB) { const float4 v = partials[r]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    a.x = wave_sum(a.x); a.y = wave_sum(a.y); a.z = wave_sum(a.z); a.w = wave_sum(a.w);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a.x; red[threadIdx.x >> 6][1] = a.y; red[threadIdx.x >> 6][2] = a.z; red[threadIdx.x >> 6][3] = a.w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4];
        for (int c = 0; c < 4; ++c) t[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        const float tss = t[3];
        const float cls = w_cls * (t[0] / tss);                                  // loss.py:162-163: divided even when the sum is 0 (inf)
        const float iou = tss > 0.f ? w_iou * (t[1] / tss) : 0.f, dfl = tss > 0.f ? w_dfl * (t[2] / tss) : 0.f;   // no foreground: BboxLoss returns zeros (:262-266)
        out[0] = cls + iou + dfl; out[1] = iou; out[2] = dfl; out[3] = cls; out[4] = tss;
    }
}

// ---- box terms: one thread per (anchor, side); waves without a foreground anchor do no loads beyond the assignment
template <typename T, bool GRAD>
__global__ __launch_bounds__(kTB) void loss_box_kernel(const T* __restrict__ distri, const float* __restrict__ pts, const float* __restrict__ st,
                                                       const float* __restrict__ gts, const int* __restrict__ agt, const float* __restrict__ norm,
                                                       int NA, int A, float w_iou, float w_dfl, const float* __restrict__ upstream,
                                                       const float* __restrict__ fwd_out, float4* __restrict__ partials, T* __restrict__ grad) {
    __shared__ float red[kTB / 64][3];
    const int tid = threadIdx.x, lane = tid & 63, side = tid & 3;
    float l_iou = 0.f, l_dfl = 0.f, l_tss = 0.f;
    for (int tile = blockIdx.x; tile * 64 < NA; tile += gridDim.x) {
    const long long an = (long long)tile * 64 + (tid >> 2);
    const int gi = an < NA ? agt[an] : -1;
    if (__any(gi >= 0)) {
        const bool fg = gi >= 0;
        const long long ac = fg ? an : 0;                                        // background lanes compute on anchor 0 and drop the result
        const int ai = (int)(ac % A);
        const float s = st[ai], px = pts[2 * ai] / s, py = pts[2 * ai + 1] / s;
        const T* z = distri + ((size_t)ac * 4 + side) * kR1;
        float e[kR1], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < kR1; ++k) { e[k] = (float)z[k]; m = fmaxf(m, e[k]); }
        float sum = 0.f, acc = 0.f;
#pragma unroll
        for (int k = 0; k < kR1; ++k) { e[k] = expf(e[k] - m); sum += e[k]; acc += e[k] * (float)k; }
        const float inv = 1.f / sum, d = acc * inv;
        const int l0 = lane & ~3;
        const float d0 = __shfl(d, l0), d1 = __shfl(d, l0 + 1), d2 = __shfl(d, l0 + 2), d3 = __shfl(d, l0 + 3);
        if (fg) {
            const float* gt = gts + (size_t)gi * 5;
            const float u1 = gt[1] / s, v1 = gt[2] / s, u2 = gt[3] / s, v2 = gt[4] / s;   // loss.py:152: target boxes in stride units
            const float bw = norm[an];
            const float x1 = px - d0, y1 = py - d1, x2 = px + d2, y2 = py + d3;
            // GIoU loss, figure_iou.py IOUloss(xyxy, giou, eps = 1e-10)
            const float eps = 1e-10f;
            const float w1 = x2 - x1, h1 = y2 - y1 + eps, w2 = u2 - u1, h2 = v2 - v1 + eps;
            const float iwr = fminf(x2, u2) - fmaxf(x1, u1), ihr = fminf(y2, v2) - fmaxf(y1, v1);
            const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
            const float inter = iw * ih, uni = w1 * h1 + w2 * h2 - inter + eps;
            const float cw = fmaxf(x2, u2) - fminf(x1, u1), ch = fmaxf(y2, v2) - fminf(y1, v1);
            const float carea = cw * ch + eps;
            const float iou = inter / uni;
            if (side == 0) { l_iou += (1.f - (iou - (carea - uni) / carea)) * bw; l_tss += bw; }
            // DFL of this side (loss.py:253-267): target distance, its two neighbouring bins
            float tgt = side == 0 ? px - u1 : side == 1 ? py - v1 : side == 2 ? u2 - px : v2 - py;
            tgt = fminf(fmaxf(tgt, 0.f), (float)(kR1 - 1) - 0.01f);
            const int tl = (int)tgt;
            const float wl = (float)(tl + 1) - tgt, wr = 1.f - wl;
            float el = 0.f, er = 0.f;
#pragma unroll
            for (int k = 0; k < kR1; ++k) { el = k == tl ? e[k] : el; er = k == tl + 1 ? e[k] : er; }
            const float ce = -(wl * logf(el * inv) + wr * logf(er * inv));
            l_dfl += ce * 0.25f * bw;
            if (GRAD) {
                // derivative of the GIoU loss with respect to this side's box coordinate (sub-gradients of min / max / clamp as autograd takes them)
                const bool xs = (side & 1) == 0, lo = side < 2;
                const float q = lo ? (xs ? x1 : y1) : (xs ? x2 : y2), r = lo ? (xs ? u1 : v1) : (xs ? u2 : v2);
                const float gt_ = q > r ? 1.f : (q == r ? 0.5f : 0.f), lt_ = q < r ? 1.f : (q == r ? 0.5f : 0.f);
                const float raw = xs ? iwr : ihr;
                const float di = (lo ? -gt_ : lt_) * (raw >= 0.f ? 1.f : 0.f);     // d(iw or ih)
                const float dc1 = lo ? -lt_ : gt_;                                   // d(cw or ch)
                const float dinter = xs ? di * ih : iw * di;
                const float darea = xs ? (lo ? -h1 : h1) : (lo ? -w1 : w1);
                const float duni = darea - dinter;
                const float diou = (dinter * uni - inter * duni) / (uni * uni);
                const float dcar = xs ? dc1 * ch : cw * dc1;
                const float dL = -diou - (duni * carea - uni * dcar) / (carea * carea);
                const float dd = lo ? -dL : dL;                                      // x1 = px - d0 ... x2 = px + d2
                const float up = upstream[0] / fwd_out[4];
                const float gi_ = up * w_iou * bw * dd, gd_ = up * w_dfl * bw * 0.25f;
                T* go = grad + ((size_t)an * 4 + side) * kR1;
#pragma unroll
                for (int k = 0; k < kR1; ++k) {
                    const float p = e[k] * inv;
                    go[k] = (T)(gi_ * p * ((float)k - d) + gd_ * (p - (k == tl ? wl : 0.f) - (k == tl + 1 ? wr : 0.f)));
                }
            }
        }
    }
    }
    if (partials == nullptr) return;
    l_iou = wave_sum(l_iou); l_dfl = wave_sum(l_dfl); l_tss = wave_sum(l_tss);
    if (lane == 0) { red[tid >> 6][0] = l_iou; red[tid >> 6][1] = l_dfl; red[tid >> 6][2] = l_tss; }
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = make_float4(0.f, red[0][0] + red[1][0] + red[2][0] + red[3][0], red[0][1] + red[1][1] + red[2][1] + red[3][1],
                                                     red[0][2] + red[1][2] + red[2][2] + red[3][2]);
}

int cls_grid(long long nvec) { return (int)std::min<long long>((nvec + kTB - 1) / kTB, 1024); }

template <typename T>
int launch_terms(const void* scores, const void* distri, const float* pts, const float* st, const float* gts, const int* agt, const float* norm,
                 int B, int A, int nc, float wc, float wi, float wd, float* partials, float* out, const float* up, void* gs, void* gd, hipStream_t s) {
    const int NA = B * A;
    const bool v8 = nc % 8 == 0;
    const int nvec = (int)((long long)NA * nc / (v8 ? 8 : 1));
    const int g1 = cls_grid(nvec), g2 = std::min((NA + 63) / 64, 1024);
    float4* p1 = reinterpret_cast<float4*>(partials);
    float4* p2 = partials ? p1 + g1 : nullptr;
    const T* sc = static_cast<const T*>(scores); const T* di = static_cast<const T*>(distri);
    if (gs != nullptr) {
        const int rc = maf_check_hip(hipMemsetAsync(gd, 0, (size_t)NA * 4 * kR1 * sizeof(T), s), "loss_terms memset");   // the kernel writes the foreground rows only
        if (rc) return rc;
        if (v8) hipLaunchKernelGGL((loss_cls_kernel<T, 8, true>), dim3(g1), dim3(kTB), 0, s, sc, gts, agt, norm, nvec, nc, wc, up, out, (float4*)nullptr, static_cast<T*>(gs));
        else hipLaunchKernelGGL((loss_cls_kernel<T, 1, true>), dim3(g1), dim3(kTB), 0, s, sc, gts, agt, norm, nvec, nc, wc, up, out, (float4*)nullptr, static_cast<T*>(gs));
        hipLaunchKernelGGL((loss_box_kernel<T, true>), dim3(g2), dim3(kTB), 0, s, di, pts, st, gts, agt, norm, NA, A, wi, wd, up, out, (float4*)nullptr, static_cast<T*>(gd));
    } else {
        if (v8) hipLaunchKernelGGL((loss_cls_kernel<T, 8, false>), dim3(g1), dim3(kTB), 0, s, sc, gts, agt, norm, nvec, nc, wc, up, out, p1, static_cast<T*>(nullptr));
        else hipLaunchKernelGGL((loss_cls_kernel<T, 1, false>), dim3(g1), dim3(kTB), 0, s, sc, gts, agt, norm, nvec, nc, wc, up, out, p1, static_cast<T*>(nullptr));
        hipLaunchKernelGGL((loss_box_kernel<T, false>), dim3(g2), dim3(kTB), 0, s, di, pts, st, gts, agt, norm, NA, A, wi, wd, up, out, p2, static_cast<T*>(nullptr));
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kTB), 0, s, p1, g1 + g2, wc, wi, wd, out);
    }
    return maf_check_hip(hipGetLastError(), "loss_terms launch");
}

}  // namespace

extern "C" int64_t maf_loss_partial_rows(int32_t B, int32_t A, int32_t nc) {
    const long long NA = (long long)B * A;
    return cls_grid(NA * nc / (nc % 8 == 0 ? 8 : 1)) + std::min<long long>((NA + 63) / 64, 1024);
}

extern "C" int maf_loss_decode(const void* pred_distri, int32_t dtype, const float* anchor_points, const float* anchor_strides, int32_t B, int32_t A,
                               int32_t reg_max, float* out_boxes, maf_stream_t stream) {
    MAF_REQUIRE(pred_distri && anchor_points && anchor_strides && out_boxes, "loss_decode: null pointer");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "loss_decode: dtype must be f16 or f32");
    MAF_REQUIRE(reg_max == kR1 - 1, "loss_decode: reg_max must be 16");
    MAF_REQUIRE(B > 0 && A > 0 && (long long)B * A < (1ll << 31), "loss_decode: bad shape");
    const int NA = B * A;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL(loss_decode_kernel<_Float16>, dim3((NA + 63) / 64), dim3(kTB), 0, s, static_cast<const _Float16*>(pred_distri), anchor_points, anchor_strides, NA, A, out_boxes);
    else hipLaunchKernelGGL(loss_decode_kernel<float>, dim3((NA + 63) / 64), dim3(kTB), 0, s, static_cast<const float*>(pred_distri), anchor_points, anchor_strides, NA, A, out_boxes);
    return maf_check_hip(hipGetLastError(), "loss_decode launch");
}

extern "C" int maf_loss_terms(const void* pred_scores, const void* pred_distri, int32_t dtype, const float* anchor_points, const float* anchor_strides,
                              const float* gts, const int32_t* assigned_gt, const float* norm, int32_t B, int32_t A, int32_t nc, int32_t reg_max,
                              float w_cls, float w_iou, float w_dfl, float* partials, float* out, const float* upstream, void* grad_scores,
                              void* grad_distri, maf_stream_t stream) {
    MAF_REQUIRE(pred_scores && pred_distri && anchor_points && anchor_strides && gts && assigned_gt && norm && out, "loss_terms: null pointer");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "loss_terms: dtype must be f16 or f32");
    MAF_REQUIRE(reg_max == kR1 - 1, "loss_terms: reg_max must be 16");
    MAF_REQUIRE(B > 0 && A > 0 && nc > 0 && (long long)B * A * nc < (1ll << 31), "loss_terms: bad shape (B*A*nc must stay below 2^31)");
    MAF_REQUIRE((grad_scores == nullptr) == (grad_distri == nullptr) && (grad_scores == nullptr) == (upstream == nullptr),
                "loss_terms: the gradient form takes upstream + both gradient buffers, the forward form none of them");
    MAF_REQUIRE(grad_scores != nullptr || partials != nullptr, "loss_terms: the forward form needs the partials scratch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) return launch_terms<_Float16>(pred_scores, pred_distri, anchor_points, anchor_strides, gts, assigned_gt, norm, B, A, nc, w_cls, w_iou, w_dfl, partials, out, upstream, grad_scores, grad_distri, s);
    return launch_terms<float>(pred_scores, pred_distri, anchor_points, anchor_strides, gts, assigned_gt, norm, B, A, nc, w_cls, w_iou, w_dfl, partials, out, upstream, grad_scores, grad_distri, s);
}
