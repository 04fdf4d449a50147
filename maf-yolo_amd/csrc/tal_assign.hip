// Task-aligned label assignment on the device, ragged (no padding to the image with most boxes, no dense [B, n_max, A] masks, no
// host round trip).  Replaces TaskAlignedAssigner.forward (yolov6/assigners/tal_assigner.py:21-151) with select_candidates_in_gts /
// select_highest_overlaps / iou_calculator (assigner_utils.py:25-89) as called from ComputeLoss.__call__ (yolov6/models/loss.py:96-103),
// and makes ComputeLoss.preprocess (:179-188: python lists + targets.cpu().numpy()) unnecessary.
//
// One workgroup (1024 threads) per image walks its ground-truth boxes:
//   per box g:   IoU with every predicted box, metric = score[label_g]^alpha * IoU^beta, "anchor centre inside the box" test;
//                metric * inside goes to an LDS row; 13 rounds of workgroup arg-max pick the top-k anchors of the row
//                (ties: lowest anchor index); picked anchors that lie inside the box count the box as a candidate;
//                every anchor also tracks the box with the largest IoU over ALL boxes (first maximum), the rule for anchors
//                picked by several boxes (assigner_utils.py:58-64);
//   resolve:     anchor -> its single candidate box, or the max-IoU box if it has several; background otherwise;
//   normalise:   per box the maxima of metric and IoU over its final anchors (LDS float-max atomics), then
//                norm[a] = metric * max_iou / (max_metric + eps)   (tal_assigner.py:66-71).
// Outputs per anchor: the index of the assigned box in the (image-sorted) target list or -1, and norm.  Labels, boxes and the
// one-hot score targets are gathers of those on the host side (maf-yolo_amd/loss.py).
#include "maf_common.h"

namespace {

constexpr int kT = 1024;
constexpr int kMaxA = 8400;           // anchors of a 640 x 640 image (3 levels); larger images are rejected by the host entry

struct TalArgs {
    const float* scores;      // [B,A,nc]
    const float* boxes;       // [B,A,4] xyxy pixels
    const float* points;      // [A,2] anchor centres, pixels
    const float* gts;         // [T,5] label, x1,y1,x2,y2 pixels, sorted by image
    const int* offs;          // [B+1] first box of every image
    int* out_gt;              // [B,A]
    float* out_norm;          // [B,A]
    int A, nc, topk;
    float alpha, beta, eps;
};

__device__ __forceinline__ float iou_box(float gx1, float gy1, float gx2, float gy2, const float4 p, float eps) {
    const float ix = fmaxf(fminf(gx2, p.z) - fmaxf(gx1, p.x), 0.f), iy = fmaxf(fminf(gy2, p.w) - fmaxf(gy1, p.y), 0.f);
    const float inter = ix * iy;
    const float a1 = fmaxf(gx2 - gx1, 0.f) * fmaxf(gy2 - gy1, 0.f), a2 = fmaxf(p.z - p.x, 0.f) * fmaxf(p.w - p.y, 0.f);
    return inter / (a1 + a2 - inter + eps);
}

__global__ __launch_bounds__(kT) void tal_assign_kernel(const TalArgs a) {
    __shared__ float row[kMaxA];                 // metric * inside of the current box
    __shared__ float best_ov[kMaxA];
    __shared__ short best_g[kMaxA], sel_g[kMaxA];
    __shared__ unsigned char cnt[kMaxA];
    __shared__ float red_v[kT / 64];
    __shared__ int red_i[kT / 64];
    __shared__ int s_pick;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = a.offs[b], n = a.offs[b + 1] - g0;
    const float* sc = a.scores + (size_t)b * a.A * a.nc;
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.A;
    for (int i = tid; i < a.A; i += kT) { best_ov[i] = -1.f; best_g[i] = 0; sel_g[i] = -1; cnt[i] = 0; }
    __syncthreads();
    for (int g = 0; g < n; ++g) {
        const float* gt = a.gts + (size_t)(g0 + g) * 5;
        const int label = (int)gt[0];
        const float gx1 = gt[1], gy1 = gt[2], gx2 = gt[3], gy2 = gt[4];
        for (int i = tid; i < a.A; i += kT) {
            const float ov = iou_box(gx1, gy1, gx2, gy2, bx[i], a.eps);
            const float px = a.points[2 * i], py = a.points[2 * i + 1];
            const float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
            const float s = sc[(size_t)i * a.nc + label];
            const float metric = powf(s, a.alpha) * powf(ov, a.beta);
            row[i] = dmin > a.eps ? metric : 0.f;
            if (ov > best_ov[i]) { best_ov[i] = ov; best_g[i] = (short)g; }     // first maximum over the boxes
        }
        __syncthreads();
        for (int k = 0; k < a.topk; ++k) {                                     // torch.topk(metric * inside, 13): one arg-max per round
            float bv = -1.f; int bi = 0x7fffffff;
            for (int i = tid; i < a.A; i += kT) {
                const float v = row[i];
                if (v > bv) { bv = v; bi = i; }                                   // strictly greater: lowest index among equals in this lane
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                float fv = red_v[0]; int fi = red_i[0];
                for (int w = 1; w < kT / 64; ++w)
                    if (red_v[w] > fv || (red_v[w] == fv && red_i[w] < fi)) { fv = red_v[w]; fi = red_i[w]; }
                s_pick = fi;
                // inside the box?  (row holds metric * inside, which is 0 for inside anchors with zero metric too: test again)
                const float px = a.points[2 * fi], py = a.points[2 * fi + 1];
                const float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
                if (dmin > a.eps) { cnt[fi] = (unsigned char)min(255, (int)cnt[fi] + 1); sel_g[fi] = (short)g; }
                row[fi] = -2.f;                                                  // out of the following rounds
            }
            __syncthreads();
        }
    }
    // ---- resolve + per-box maxima (row / best_ov reused as the per-box float-max accumulators: n <= kMaxA boxes)
    __syncthreads();
    int my_gt[(kMaxA + kT - 1) / kT];
    float my_m[(kMaxA + kT - 1) / kT], my_o[(kMaxA + kT - 1) / kT];
    unsigned int* max_m = reinterpret_cast<unsigned int*>(row);
    unsigned int* max_o = reinterpret_cast<unsigned int*>(best_ov);
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kT, ++u) {
            const int c = cnt[i];
            my_gt[u] = c == 0 ? -1 : (c == 1 ? (int)sel_g[i] : (int)best_g[i]);
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += kT) { max_m[i] = 0u; max_o[i] = 0u; }
    __syncthreads();
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kT, ++u) {
            const int g = my_gt[u];
            if (g < 0) continue;
            const float* gt = a.gts + (size_t)(g0 + g) * 5;
            const float ov = iou_box(gt[1], gt[2], gt[3], gt[4], bx[i], a.eps);
            const float m = powf(sc[(size_t)i * a.nc + (int)gt[0]], a.alpha) * powf(ov, a.beta);
            my_m[u] = m; my_o[u] = ov;
            atomicMax(&max_m[g], __float_as_uint(m));                           // non-negative floats order like their bit patterns
            atomicMax(&max_o[g], __float_as_uint(ov));
        }
    }
    __syncthreads();
    {
        int u = 0;
        for (int i = tid; i < a.A; i += kT, ++u) {
            const int g = my_gt[u];
            a.out_gt[(size_t)b * a.A + i] = g < 0 ? -1 : g0 + g;
            a.out_norm[(size_t)b * a.A + i] = g < 0 ? 0.f : my_m[u] * __uint_as_float(max_o[g]) / (__uint_as_float(max_m[g]) + a.eps);
        }
    }
}

}  // namespace

extern "C" int maf_tal_assign(const float* pd_scores, const float* pd_bboxes, const float* anchor_points, const float* gts, const int32_t* offsets,
                              int32_t B, int32_t A, int32_t nc, int32_t topk, float alpha, float beta, float eps,
                              int32_t* out_gt, float* out_norm, maf_stream_t stream) {
    MAF_REQUIRE(pd_scores && pd_bboxes && anchor_points && gts && offsets && out_gt && out_norm, "tal_assign: null pointer");
    MAF_REQUIRE(B > 0 && A > 0 && A <= kMaxA && nc > 0 && topk > 0 && topk <= A, "tal_assign: bad shape (at most 8400 anchors per image)");
    TalArgs a;
    a.scores = pd_scores; a.boxes = pd_bboxes; a.points = anchor_points; a.gts = gts; a.offs = offsets; a.out_gt = out_gt; a.out_norm = out_norm;
    a.A = A; a.nc = nc; a.topk = topk; a.alpha = alpha; a.beta = beta; a.eps = eps;
    hipLaunchKernelGGL(tal_assign_kernel, dim3(B), dim3(kT), 0, static_cast<hipStream_t>(stream), a);
    return maf_check_hip(hipGetLastError(), "tal_assign launch");
}
