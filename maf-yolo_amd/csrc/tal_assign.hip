// Task-aligned label assignment on the device, ragged (no padding to the image with most boxes, no dense [B, n_max, A] masks, no
// host round trip).  Replaces TaskAlignedAssigner.forward (yolov6/assigners/tal_assigner.py:21-151) with select_candidates_in_gts /
// select_highest_overlaps / iou_calculator (assigner_utils.py:25-89) as called from ComputeLoss.__call__ (yolov6/models/loss.py:96-103),
// and makes ComputeLoss.preprocess (:179-188: python lists + targets.cpu().numpy()) unnecessary.
//
// Two launches:
//   tal_topk_kernel     one workgroup (256 threads) per ground-truth box, all boxes of the batch in parallel:
//                       metric = score[label]^alpha * IoU^beta for the anchors whose centre lies inside the box (only those load a
//                       predicted box and gather a score; every other anchor is -0.0, which ties with a zero metric but keeps
//                       "outside" in the sign bit), then 13 rounds of workgroup arg-max (ties: lowest anchor index, one barrier per
//                       round) -> cand[box][k] = anchor, or ~anchor if the pick lies outside the box (not a candidate).
//   tal_resolve_kernel  one workgroup (1024 threads) per image: count the boxes that picked every anchor (LDS atomics);
//                       one box -> assigned, several -> the box with the largest IoU over ALL boxes of the image (first maximum,
//                       assigner_utils.py:58-64), none -> background; then per box the maxima of metric and IoU over its final
//                       anchors (LDS float-max atomics) and norm[a] = metric * max_iou / (max_metric + eps) (tal_assigner.py:66-71).
// Outputs per anchor: the row of the assigned box in the (image-sorted) target list or -1, and norm.  Labels, boxes and the one-hot
// score targets are never materialised: the loss kernels (loss_terms.hip) read these two arrays.
#include "maf_common.h"

namespace {

constexpr int kTK = 256;              // threads of the per-box kernel
constexpr int kTR = 1024;             // threads of the per-image kernel
constexpr int kMaxA = 8400;           // anchors of a 640 x 640 image (3 levels); larger images are rejected by the host entry

struct TalArgs {
    const void* scores;       // [B,A,nc] f16 or f32
    const float* boxes;       // [B,A,4] xyxy pixels
    const float* points;      // [A,2] anchor centres, pixels
    const float* gts;         // [T,5] label, x1,y1,x2,y2 pixels, sorted by image
    const int* gt_img;        // [T] image of every box
    const int* offs;          // [B+1] first box of every image
    int* cand;                // [T,topk]
    int* out_gt;              // [B,A]
    float* out_norm;          // [B,A]
    int A, nc, topk;
    float alpha, beta, eps;
};

template <typename T> __device__ __forceinline__ float score_at(const void* p, size_t i) { return (float)static_cast<const T*>(p)[i]; }

__device__ __forceinline__ float iou_box(float gx1, float gy1, float gx2, float gy2, const float4 p, float eps) {
    const float ix = fmaxf(fminf(gx2, p.z) - fmaxf(gx1, p.x), 0.f), iy = fmaxf(fminf(gy2, p.w) - fmaxf(gy1, p.y), 0.f);
    const float inter = ix * iy;
    const float a1 = fmaxf(gx2 - gx1, 0.f) * fmaxf(gy2 - gy1, 0.f), a2 = fmaxf(p.z - p.x, 0.f) * fmaxf(p.w - p.y, 0.f);
    return inter / (a1 + a2 - inter + eps);
}

template <typename T>
__global__ __launch_bounds__(kTK) void tal_topk_kernel(const TalArgs a) {
    __shared__ float row[kMaxA];                  // metric of the inside anchors, -0.0 outside
    __shared__ float red_v[2][kTK / 64];
    __shared__ int red_i[2][kTK / 64];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = a.gt_img[g];
    const float* gt = a.gts + (size_t)g * 5;
    const int label = (int)gt[0];
    const float gx1 = gt[1], gy1 = gt[2], gx2 = gt[3], gy2 = gt[4];
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.A;
    const size_t sbase = (size_t)b * a.A * a.nc + label;
    const float2* pts = reinterpret_cast<const float2*>(a.points);
    for (int i = tid; i < a.A; i += kTK) {
        const float2 p = pts[i];
        const float dmin = fminf(fminf(p.x - gx1, p.y - gy1), fminf(gx2 - p.x, gy2 - p.y));
        float v = -0.f;
        if (dmin > a.eps) {
            const float ov = iou_box(gx1, gy1, gx2, gy2, bx[i], a.eps);
            v = powf(score_at<T>(a.scores, sbase + (size_t)i * a.nc), a.alpha) * powf(ov, a.beta);
        }
        row[i] = v;
    }
    // every thread only ever reads the row entries it wrote (i = tid mod 256): no barrier needed before the rounds
    for (int k = 0; k < a.topk; ++k) {                                         // torch.topk(metric * inside, 13): one arg-max per round
        float bv = -1.f; int bi = 0x7fffffff;
        for (int i = tid; i < a.A; i += kTK) {
            const float v = row[i];
            if (v > bv) { bv = v; bi = i; }                                       // strictly greater: lowest index among equals in this lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[k & 1][wave] = bv; red_i[k & 1][wave] = bi; }
        __syncthreads();
        float fv = red_v[k & 1][0]; int fi = red_i[k & 1][0];
#pragma unroll
        for (int w = 1; w < kTK / 64; ++w) {
            const float v = red_v[k & 1][w]; const int i = red_i[k & 1][w];
            if (v > fv || (v == fv && i < fi)) { fv = v; fi = i; }
        }
        if ((fi & (kTK - 1)) == tid) {                                           // the thread that owns this row entry
            a.cand[(size_t)g * a.topk + k] = signbit(row[fi]) ? ~fi : fi;
            row[fi] = -2.f;                                                      // out of the following rounds
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kTR) void tal_resolve_kernel(const TalArgs a) {
    __shared__ int cnt[kMaxA];                    // boxes that picked the anchor; later the per-box metric maxima
    __shared__ int sel[kMaxA];                    // one of them; later the per-box IoU maxima
    const int b = blockIdx.x, tid = threadIdx.x;
    const int g0 = a.offs[b], n = min(a.offs[b + 1] - g0, kMaxA);
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.A;
    for (int i = tid; i < a.A; i += kTR) { cnt[i] = 0; sel[i] = -1; }
    __syncthreads();
    for (int e = tid; e < n * a.topk; e += kTR) {
        const int c = a.cand[(size_t)g0 * a.topk + e];
        if (c >= 0) { atomicAdd(&cnt[c], 1); sel[c] = e / a.topk; }
    }
    __syncthreads();
    constexpr int U = (kMaxA + kTR - 1) / kTR;
    int my_gt[U];
    float my_m[U], my_o[U];
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kTR, ++u) {
            const int c = cnt[i];
            int g = c == 0 ? -1 : sel[i];
            if (c > 1) {                                                          // picked by several boxes: largest IoU over all boxes, first maximum
                const float4 p = bx[i];
                float best = -1.f;
                for (int j = 0; j < n; ++j) {
                    const float* gt = a.gts + (size_t)(g0 + j) * 5;
                    const float ov = iou_box(gt[1], gt[2], gt[3], gt[4], p, a.eps);
                    if (ov > best) { best = ov; g = j; }
                }
            }
            my_gt[u] = g;
        }
    }
    __syncthreads();
    unsigned int* max_m = reinterpret_cast<unsigned int*>(cnt);
    unsigned int* max_o = reinterpret_cast<unsigned int*>(sel);
    for (int i = tid; i < n; i += kTR) { max_m[i] = 0u; max_o[i] = 0u; }
    __syncthreads();
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kTR, ++u) {
            const int g = my_gt[u];
            if (g < 0) continue;
            const float* gt = a.gts + (size_t)(g0 + g) * 5;
            const float ov = iou_box(gt[1], gt[2], gt[3], gt[4], bx[i], a.eps);
            const float m = powf(score_at<T>(a.scores, ((size_t)b * a.A + i) * a.nc + (int)gt[0]), a.alpha) * powf(ov, a.beta);
            my_m[u] = m; my_o[u] = ov;
            atomicMax(&max_m[g], __float_as_uint(m));                           // non-negative floats order like their bit patterns
            atomicMax(&max_o[g], __float_as_uint(ov));
        }
    }
    __syncthreads();
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kTR, ++u) {
            const int g = my_gt[u];
            a.out_gt[(size_t)b * a.A + i] = g < 0 ? -1 : g0 + g;
            a.out_norm[(size_t)b * a.A + i] = g < 0 ? 0.f : my_m[u] * __uint_as_float(max_o[g]) / (__uint_as_float(max_m[g]) + a.eps);
        }
    }
}

}  // namespace

extern "C" int maf_tal_assign(const void* pd_scores, int32_t score_dtype, const float* pd_bboxes, const float* anchor_points, const float* gts,
                              const int32_t* gt_image, const int32_t* offsets, int32_t T, int32_t B, int32_t A, int32_t nc, int32_t topk,
                              float alpha, float beta, float eps, int32_t* cand_scratch, int32_t* out_gt, float* out_norm, maf_stream_t stream) {
    MAF_REQUIRE(pd_scores && pd_bboxes && anchor_points && offsets && out_gt && out_norm, "tal_assign: null pointer");
    MAF_REQUIRE(T == 0 || (gts && gt_image && cand_scratch), "tal_assign: null target pointer");
    MAF_REQUIRE(score_dtype == MAF_F16 || score_dtype == MAF_F32, "tal_assign: scores must be f16 or f32");
    MAF_REQUIRE(B > 0 && T >= 0 && A > 0 && A <= kMaxA && nc > 0 && topk > 0 && topk <= A, "tal_assign: bad shape (at most 8400 anchors per image)");
    TalArgs a;
    a.scores = pd_scores; a.boxes = pd_bboxes; a.points = anchor_points; a.gts = gts; a.gt_img = gt_image; a.offs = offsets;
    a.cand = cand_scratch; a.out_gt = out_gt; a.out_norm = out_norm;
    a.A = A; a.nc = nc; a.topk = topk; a.alpha = alpha; a.beta = beta; a.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (score_dtype == MAF_F16) {
        if (T > 0) hipLaunchKernelGGL(tal_topk_kernel<_Float16>, dim3(T), dim3(kTK), 0, s, a);
        hipLaunchKernelGGL(tal_resolve_kernel<_Float16>, dim3(B), dim3(kTR), 0, s, a);
    } else {
        if (T > 0) hipLaunchKernelGGL(tal_topk_kernel<float>, dim3(T), dim3(kTK), 0, s, a);
        hipLaunchKernelGGL(tal_resolve_kernel<float>, dim3(B), dim3(kTR), 0, s, a);
    }
    return maf_check_hip(hipGetLastError(), "tal_assign launch");
}
