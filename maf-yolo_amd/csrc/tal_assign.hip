// Task-aligned label assignment on the device, ragged (no padding to the image with most boxes, no dense [B, n_max, A] masks, no
// host round trip).  Replaces TaskAlignedAssigner.forward (yolov6/assigners/tal_assigner.py:21-151) with select_candidates_in_gts /
// select_highest_overlaps / iou_calculator (assigner_utils.py:25-89) as called from ComputeLoss.__call__ (yolov6/models/loss.py:96-103),
// and makes ComputeLoss.preprocess (:179-188: python lists + targets.cpu().numpy()) unnecessary.
//
// Three launches (maf_tal_targets, then two in maf_tal_assign):
//   tal_targets_kernel  the reference's label preprocessing: rows grouped by image, xywh -> xyxy pixels, group offsets;
//   tal_topk_kernel     one workgroup (256 threads) per ground-truth box, all boxes of the batch in parallel:
//                       metric = score[label]^alpha * IoU^beta for the anchors whose centre lies inside the box — only the cell
//                       rectangle the box covers on every level is visited — kept in registers as 64-bit keys (metric | anchor),
//                       then up to 13 rounds of workgroup arg-max (ties: lowest anchor, one barrier per round); when fewer than 13
//                       metrics are positive the remaining picks are zeros in anchor order (what torch.topk returns), which count
//                       if they lie inside the box -> cand[box][k] = anchor or -1.
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
    int A, nc, topk, B;
    float alpha, beta, eps;
    int nl, lbase[4], lw[4], lh[4];   // anchor levels: first anchor, grid width / height
    float lstride[4], loff;           // anchor centre = (cell + loff) * stride
    int atss;                         // resolve kernel: 1 = ATSS rules (anchor-box IoU decides shared anchors, score target = IoU with the predicted box)
    float half_cells;                 // ATSS: anchor box = centre -/+ half_cells * stride
};

template <typename T> __device__ __forceinline__ float score_at(const void* p, size_t i) { return (float)static_cast<const T*>(p)[i]; }

// x^e for x >= 0.  The library powf costs a few hundred instructions and the assigner calls it twice per (box, inside anchor); the
// exponents of the reference are alpha = 1 and beta = 6 (loss.py:46): e = 1 is the identity and a small whole e is a handful of double
// multiplications rounded once to float (within half an ulp of the exact power, like a correctly rounded pow).  Anything else: powf.
__device__ __attribute__((noinline)) float pow_slow(float x, float e) { return powf(x, e); }   // one copy: inlined it is 300 instructions per call site

__device__ __forceinline__ float pow_pos(float x, float e) {
    if (e == 1.f) return x;
    const int n = (int)e;
    if ((float)n == e && n >= 2 && n <= 16) {
        double r = 1.0, b = (double)x;
        for (int k = n; k > 0; k >>= 1) { if (k & 1) r *= b; b *= b; }
        return (float)r;
    }
    return pow_slow(x, e);
}

__device__ __forceinline__ float iou_box(float gx1, float gy1, float gx2, float gy2, const float4 p, float eps) {
    const float ix = fmaxf(fminf(gx2, p.z) - fmaxf(gx1, p.x), 0.f), iy = fmaxf(fminf(gy2, p.w) - fmaxf(gy1, p.y), 0.f);
    const float inter = ix * iy;
    const float a1 = fmaxf(gx2 - gx1, 0.f) * fmaxf(gy2 - gy1, 0.f), a2 = fmaxf(p.z - p.x, 0.f) * fmaxf(p.w - p.y, 0.f);
    return inter / (a1 + a2 - inter + eps);
}

// the square anchor box of anchor i (anchor_generator.py:29-38) and IoU as iou2d_calculator.bbox_overlaps computes it (eps 1e-6 as a floor of the union)
__device__ __forceinline__ float4 anchor_box(const TalArgs& a, int i) {
    const int l = (i >= a.lbase[1] && a.nl > 1) + (i >= a.lbase[2] && a.nl > 2) + (i >= a.lbase[3] && a.nl > 3);
    const float s = l == 0 ? a.lstride[0] : l == 1 ? a.lstride[1] : l == 2 ? a.lstride[2] : a.lstride[3];
    const float2 p = reinterpret_cast<const float2*>(a.points)[i];
    const float h = a.half_cells * s;
    return make_float4(p.x - h, p.y - h, p.x + h, p.y + h);
}
__device__ __forceinline__ float iou_anchor(float gx1, float gy1, float gx2, float gy2, const float4 p) {
    const float a1 = (gx2 - gx1) * (gy2 - gy1), a2 = (p.z - p.x) * (p.w - p.y);
    const float w = fmaxf(fminf(gx2, p.z) - fmaxf(gx1, p.x), 0.f), h = fmaxf(fminf(gy2, p.w) - fmaxf(gy1, p.y), 0.f);
    const float inter = w * h;
    return inter / fmaxf(a1 + a2 - inter, 1e-6f);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

// key of an anchor in the top-k rounds: metric bits (non-negative float: orders like its bit pattern) | 16383 - anchor (lowest anchor
// wins among equal metrics) | "centre inside the box" in bit 0.  0 = taken / no anchor.
__device__ __forceinline__ unsigned long long topk_key(float v, int i, bool inside) {
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(((16383 - i) << 1) | (inside ? 1 : 0));
}

template <typename T>
__global__ __launch_bounds__(kTK) void tal_topk_kernel(const TalArgs a) {
    __shared__ unsigned long long red[2][kTK / 64];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (g >= a.offs[a.B]) return;                                              // rows the target preprocessing dropped (image id out of range)
    const int b = a.gt_img[g];
    const float* gt = a.gts + (size_t)g * 5;
    const int label = (int)gt[0];
    const float gx1 = gt[1], gy1 = gt[2], gx2 = gt[3], gy2 = gt[4];
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.A;
    const size_t sbase = (size_t)b * a.A * a.nc + label;
    const float2* pts = reinterpret_cast<const float2*>(a.points);
    if (tid < a.topk) a.cand[(size_t)g * a.topk + tid] = -1;                    // "picked nothing that lies inside"
    // Only anchors whose centre lies inside the box can have a positive metric: enumerate the cell rectangle the box covers on every
    // level (one cell of slack; the exact test below uses the anchor_points array) instead of all A anchors.
    int lx0[4], lwd[4], ly0[4], lcum[5];
    lcum[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        int n = 0;
        lx0[l] = 0; ly0[l] = 0; lwd[l] = 1;
        if (l < a.nl) {
            const float inv = 1.f / a.lstride[l];
            const int x0 = max((int)floorf(gx1 * inv - a.loff) - 1, 0), x1 = min((int)ceilf(gx2 * inv - a.loff) + 1, a.lw[l] - 1);
            const int y0 = max((int)floorf(gy1 * inv - a.loff) - 1, 0), y1 = min((int)ceilf(gy2 * inv - a.loff) + 1, a.lh[l] - 1);
            if (x1 >= x0 && y1 >= y0) { lx0[l] = x0; ly0[l] = y0; lwd[l] = x1 - x0 + 1; n = lwd[l] * (y1 - y0 + 1); }
        }
        lcum[l + 1] = lcum[l] + n;
    }
    const int N = lcum[4];
    constexpr int U = (kMaxA + kTK - 1) / kTK, CH = 11;                        // up to 33 anchors per thread, built 11 at a time
    static_assert(U % CH == 0, "chunking");
    unsigned long long key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = 0ull;
#pragma unroll
    for (int c = 0; c < U; c += CH) {
        if (c * kTK < N) {
            // loads first and unconditional (slots past the rectangle and anchors outside the box read anchor 0: one cached line), arithmetic after
            int ai[CH]; float2 p[CH]; float4 bb[CH]; float ss[CH]; bool in[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = tid + (c + u) * kTK;
                const int l = (j >= lcum[1]) + (j >= lcum[2]) + (j >= lcum[3]);
                const int r = j - (l == 0 ? lcum[0] : l == 1 ? lcum[1] : l == 2 ? lcum[2] : lcum[3]);
                const int wd = l == 0 ? lwd[0] : l == 1 ? lwd[1] : l == 2 ? lwd[2] : lwd[3];
                const int dy = r / wd, dx = r - dy * wd;
                const int xx = (l == 0 ? lx0[0] : l == 1 ? lx0[1] : l == 2 ? lx0[2] : lx0[3]) + dx;
                const int yy = (l == 0 ? ly0[0] : l == 1 ? ly0[1] : l == 2 ? ly0[2] : ly0[3]) + dy;
                const int lb = l == 0 ? a.lbase[0] : l == 1 ? a.lbase[1] : l == 2 ? a.lbase[2] : a.lbase[3];
                const int lw_ = l == 0 ? a.lw[0] : l == 1 ? a.lw[1] : l == 2 ? a.lw[2] : a.lw[3];
                ai[u] = j < N ? lb + yy * lw_ + xx : -1;
                p[u] = pts[max(ai[u], 0)];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float dmin = fminf(fminf(p[u].x - gx1, p[u].y - gy1), fminf(gx2 - p[u].x, gy2 - p[u].y));
                in[u] = ai[u] >= 0 && dmin > a.eps;
                const int idx = in[u] ? ai[u] : 0;
                bb[u] = bx[idx];
                ss[u] = score_at<T>(a.scores, sbase + (size_t)idx * a.nc);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                float v = 0.f;
                if (in[u]) v = pow_pos(ss[u], a.alpha) * pow_pos(iou_box(gx1, gy1, gx2, gy2, bb[u], a.eps), a.beta);
                key[c + u] = v > 0.f ? topk_key(v, ai[u], true) : 0ull;            // zero metrics are settled by anchor order below
            }
        }
    }
    unsigned long long lmax = 0ull;
#pragma unroll
    for (int u = 0; u < U; ++u) lmax = key[u] > lmax ? key[u] : lmax;
    __syncthreads();                                                            // the -1 fill above, before anyone writes a pick
    int k = 0;
    for (; k < a.topk; ++k) {                                                  // torch.topk(metric * inside, 13): one arg-max per round, one barrier
        const unsigned long long m = wave_max_u64(lmax);
        if (lane == 0) red[k & 1][wave] = m;
        __syncthreads();
        unsigned long long f = red[k & 1][0];
#pragma unroll
        for (int w = 1; w < kTK / 64; ++w) f = red[k & 1][w] > f ? red[k & 1][w] : f;
        if (f == 0ull) break;                                                   // no positive metric left (uniform)
        if (lmax == f) {                                                        // keys are unique: exactly one thread owns the pick
            a.cand[(size_t)g * a.topk + k] = 16383 - (int)(((unsigned)f >> 1) & 16383u);
            lmax = 0ull;
#pragma unroll
            for (int u = 0; u < U; ++u) { key[u] = key[u] == f ? 0ull : key[u]; lmax = key[u] > lmax ? key[u] : lmax; }
        }
    }
    if (k < a.topk && wave == 0) {
        // fewer than top-k positive metrics: the rest of the picks are zeros, which top-k takes in anchor order from anchor 0 — at most
        // topk - 1 of the first 64 anchors are positive, so one wave sees enough of them.  A zero pick counts if it lies inside the box.
        const int i = lane;
        bool inside = false, positive = false;
        if (i < a.A) {
            const float2 p = pts[i];
            inside = fminf(fminf(p.x - gx1, p.y - gy1), fminf(gx2 - p.x, gy2 - p.y)) > a.eps;
            if (inside) positive = pow_pos(score_at<T>(a.scores, sbase + (size_t)i * a.nc), a.alpha) * pow_pos(iou_box(gx1, gy1, gx2, gy2, bx[i], a.eps), a.beta) > 0.f;
        }
        const bool zero = i < a.A && !positive;
        const unsigned long long mask = __ballot(zero);
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (zero && inside && rank < a.topk - k) a.cand[(size_t)g * a.topk + k + rank] = i;
    }
}

constexpr int kGtL = 1024;           // boxes of an image staged in LDS by the resolve kernel (the rest is read from memory)
constexpr int kMulti = 4096;          // anchors picked by several boxes that are queued for the balanced pass

template <typename T>
__global__ __launch_bounds__(kTR) void tal_resolve_kernel(const TalArgs a) {
    __shared__ int cnt[kMaxA];                    // boxes that picked the anchor; later the per-box metric maxima
    __shared__ int sel[kMaxA];                    // one of them (or the max-IoU box); later the per-box IoU maxima
    __shared__ float gl[kGtL * 5];
    __shared__ int mlist[kMulti];
    __shared__ int n_multi;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int g0 = a.offs[b], n = min(a.offs[b + 1] - g0, kMaxA);
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.A;
    for (int i = tid; i < a.A; i += kTR) { cnt[i] = 0; sel[i] = -1; }
    for (int i = tid; i < min(n, kGtL) * 5; i += kTR) gl[i] = a.gts[(size_t)g0 * 5 + i];
    if (tid == 0) n_multi = 0;
    __syncthreads();
    for (int e = tid; e < n * a.topk; e += kTR) {
        const int c = a.cand[(size_t)g0 * a.topk + e];
        if (c >= 0) { atomicAdd(&cnt[c], 1); sel[c] = e / a.topk; }
    }
    __syncthreads();
    // anchors picked by several boxes go to the box with the largest IoU over ALL boxes of the image (first maximum).  Queue them and
    // give every thread one: they are a few per cent of the anchors, scattered over all waves.
    auto best_box = [&](int i) {
        float4 p = bx[i];
        if (a.atss) p = anchor_box(a, i);
        float best = -1.f; int g = 0;
        for (int j = 0; j < n; ++j) {
            const float* gt = j < kGtL ? gl + j * 5 : a.gts + (size_t)(g0 + j) * 5;
            const float ov = a.atss ? iou_anchor(gt[1], gt[2], gt[3], gt[4], p) : iou_box(gt[1], gt[2], gt[3], gt[4], p, a.eps);
            if (ov > best) { best = ov; g = j; }
        }
        return g;
    };
    for (int i = tid; i < a.A; i += kTR)
        if (cnt[i] > 1) {
            const int q = atomicAdd(&n_multi, 1);
            if (q < kMulti) mlist[q] = i; else sel[i] = best_box(i);
        }
    __syncthreads();
    for (int e = tid; e < min(n_multi, kMulti); e += kTR) { const int i = mlist[e]; sel[i] = best_box(i); }
    __syncthreads();
    constexpr int U = (kMaxA + kTR - 1) / kTR;
    int my_gt[U];
    float my_m[U], my_o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = tid + u * kTR;
        my_gt[u] = i < a.A && cnt[i] > 0 ? sel[i] : -1;
    }
    __syncthreads();
    unsigned int* max_m = reinterpret_cast<unsigned int*>(cnt);
    unsigned int* max_o = reinterpret_cast<unsigned int*>(sel);
    for (int i = tid; i < n; i += kTR) { max_m[i] = 0u; max_o[i] = 0u; }
    // operands of every assigned anchor: unconditional loads (background anchors read row g0 / anchor 0), then the arithmetic
    float4 bb[U]; float ss[U]; float gq[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool on = my_gt[u] >= 0;
        const int i = on ? tid + u * kTR : 0;
        const float* gt = a.gts + (size_t)(g0 + (on ? my_gt[u] : 0)) * 5;
        const float lab = n > 0 ? gt[0] : 0.f;
        gq[u][0] = gt[1]; gq[u][1] = gt[2]; gq[u][2] = gt[3]; gq[u][3] = gt[4];
        bb[u] = bx[i];
        ss[u] = a.atss ? 1.f : score_at<T>(a.scores, ((size_t)b * a.A + i) * a.nc + (on ? (int)lab : 0));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int g = my_gt[u];
        if (g < 0) continue;
        const float ov = iou_box(gq[u][0], gq[u][1], gq[u][2], gq[u][3], bb[u], a.eps);
        const float m = a.atss ? ov : pow_pos(ss[u], a.alpha) * pow_pos(ov, a.beta);
        my_m[u] = m; my_o[u] = ov;
        atomicMax(&max_m[g], __float_as_uint(m));                               // non-negative floats order like their bit patterns
        atomicMax(&max_o[g], __float_as_uint(ov));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = tid + u * kTR, g = my_gt[u];
        if (i >= a.A) continue;
        a.out_gt[(size_t)b * a.A + i] = g < 0 ? -1 : g0 + g;
        a.out_norm[(size_t)b * a.A + i] = g < 0 ? 0.f : a.atss ? my_o[u] : my_m[u] * __uint_as_float(max_o[g]) / (__uint_as_float(max_m[g]) + a.eps);
    }
}

// ATSS candidates (yolov6/assigners/atss_assigner.py:55-74, 89-137): per box the 9 anchors of every level nearest to its centre, the
// threshold mean + std of their anchor-box IoUs, positives = candidates above it whose centre lies inside the box.  One workgroup per
// box; the 9 nearest grid points of a level lie within 4 cells of the cell that holds the box centre, so a 9 x 9 window per level
// (one thread per cell, 243 of the 256) replaces the reference's distance matrix and top-k over all anchors.  Equal distances: lowest
// anchor first (torch.topk leaves that order open).
constexpr int kAW = 9, kAS = kAW * kAW;
__global__ __launch_bounds__(kTK) void atss_cand_kernel(const TalArgs a) {
    __shared__ float sd[3 * kAS];
    __shared__ int si[3 * kAS];
    __shared__ float cov[32];
    __shared__ float s_thr;
    const int g = blockIdx.x, tid = threadIdx.x;
    if (g >= a.offs[a.B]) return;
    const float* gt = a.gts + (size_t)g * 5;
    const float gx1 = gt[1], gy1 = gt[2], gx2 = gt[3], gy2 = gt[4];
    const bool gvalid = gx1 + gy1 + gx2 + gy2 > 0.f;                            // loss.py:77 mask_gt
    const float gcx = (gx1 + gx2) / 2.0f, gcy = (gy1 + gy2) / 2.0f;
    const int l = tid / kAS, slot = tid - l * kAS;
    int i = -1; float d = 0.f; float4 ab = make_float4(0.f, 0.f, 0.f, 0.f); float acx = 0.f, acy = 0.f;
    if (l < a.nl) {
        const float s = a.lstride[l];
        const int cx = min(max((int)floorf(gcx / s - a.loff + 0.5f), 0), a.lw[l] - 1), cy = min(max((int)floorf(gcy / s - a.loff + 0.5f), 0), a.lh[l] - 1);
        const int x = cx + slot % kAW - kAW / 2, y = cy + slot / kAW - kAW / 2;
        if (x >= 0 && x < a.lw[l] && y >= 0 && y < a.lh[l]) {
            i = a.lbase[l] + y * a.lw[l] + x;
            ab = anchor_box(a, i);
            acx = (ab.x + ab.z) / 2.0f; acy = (ab.y + ab.w) / 2.0f;               // assigner_utils.py:17-20
            const float dx = gcx - acx, dy = gcy - acy;
            d = sqrtf(dx * dx + dy * dy);
        }
    }
    if (tid < 3 * kAS) { sd[tid] = d; si[tid] = i; }
    __syncthreads();
    int pos = -1;
    if (i >= 0) {
        int rank = 0;
        for (int q = l * kAS; q < (l + 1) * kAS; ++q) {
            const int j = si[q];
            rank += (j >= 0 && (sd[q] < d || (sd[q] == d && j < i))) ? 1 : 0;
        }
        if (rank < a.topk) pos = l * a.topk + rank;
    }
    const float ov = pos >= 0 ? iou_anchor(gx1, gy1, gx2, gy2, ab) : 0.f;
    if (pos >= 0) cov[pos] = ov;
    __syncthreads();
    if (tid == 0) {
        const int nc_ = a.nl * a.topk;
        float sum = 0.f;
        for (int k = 0; k < nc_; ++k) sum += cov[k];
        const float mean = sum / (float)nc_;
        float var = 0.f;
        for (int k = 0; k < nc_; ++k) var += (cov[k] - mean) * (cov[k] - mean);
        s_thr = mean + sqrtf(var / (float)(nc_ - 1));                              // torch.std: unbiased
    }
    __syncthreads();
    if (pos >= 0) {
        const float dmin = fminf(fminf(acx - gx1, acy - gy1), fminf(gx2 - acx, gy2 - acy));
        a.cand[(size_t)g * a.nl * a.topk + pos] = (ov > s_thr && dmin > 1e-9f && gvalid) ? i : -1;
    }
}

// ComputeLoss.preprocess (yolov6/models/loss.py:179-188) on the device: labels [T,6] = (image, class, cx, cy, w, h) normalised -> rows grouped
// by image in their original order, boxes as xyxy pixels, offsets of the groups.  One workgroup; rank of a row inside its image by a
// broadcast scan of the image ids in LDS (T^2 / 1024 LDS reads per thread: microseconds for a batch's few hundred to few thousand labels).
constexpr int kMaxT = 16384, kMaxB = 4096;
__global__ __launch_bounds__(kTR) void tal_targets_kernel(const float* __restrict__ tg, int T, int B, float img_size, float* __restrict__ gts,
                                                          int* __restrict__ gt_img, int* __restrict__ offs) {
    __shared__ short ids[kMaxT];
    __shared__ int start[kMaxB + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i <= B; i += kTR) start[i] = 0;
    __syncthreads();
    for (int r = tid; r < T; r += kTR) {
        const float f = tg[(size_t)r * 6];
        const int im = (f >= 0.f && f < (float)B) ? (int)f : -1;               // rows of images outside the batch are dropped
        ids[r] = (short)im;
        if (im >= 0) atomicAdd(&start[im + 1], 1);
    }
    __syncthreads();
    if (tid == 0) for (int i = 0; i < B; ++i) start[i + 1] += start[i];
    __syncthreads();
    for (int i = tid; i <= B; i += kTR) offs[i] = start[i];
    // a collate function that concatenates the labels image by image hands them over already grouped: then row r stays row r
    int bad = 0;
    for (int r = tid; r < T; r += kTR) bad |= ids[r] < 0 || (r > 0 && ids[r - 1] > ids[r]);
    const bool grouped = __syncthreads_or(bad) == 0;
    for (int r = tid; r < T; r += kTR) {
        const int im = ids[r];
        if (im < 0) continue;
        int pos = r;
        if (!grouped) {
            int rank = 0;
#pragma unroll 8
            for (int q = 0; q < r; ++q) rank += ids[q] == im ? 1 : 0;
            pos = start[im] + rank;
        }
        const float* t = tg + (size_t)r * 6;
        const float cx = t[2] * img_size, cy = t[3] * img_size, w = t[4] * img_size, h = t[5] * img_size;   // loss.py:186-187 (scale, then xywh2xyxy)
        float* o = gts + (size_t)pos * 5;
        o[0] = t[1]; o[1] = cx - w / 2; o[2] = cy - h / 2; o[3] = cx + w / 2; o[4] = cy + h / 2;
        gt_img[pos] = im;
    }
}

}  // namespace

static int set_levels(TalArgs& a, int A, int n_levels, const int32_t* level_hw, const float* level_stride, float cell_offset) {
    MAF_REQUIRE(n_levels >= 1 && n_levels <= 4 && level_hw && level_stride, "assign: 1..4 anchor levels");
    int base = 0;
    for (int l = 0; l < 4; ++l) {
        a.lbase[l] = base; a.lh[l] = l < n_levels ? level_hw[2 * l] : 0; a.lw[l] = l < n_levels ? level_hw[2 * l + 1] : 0;
        a.lstride[l] = l < n_levels ? level_stride[l] : 1.f;
        MAF_REQUIRE(l >= n_levels || (a.lh[l] > 0 && a.lw[l] > 0 && a.lstride[l] > 0.f), "assign: bad level");
        base += a.lh[l] * a.lw[l];
    }
    MAF_REQUIRE(base == A, "assign: the levels must add up to A anchors");
    a.nl = n_levels; a.loff = cell_offset;
    return 0;
}

extern "C" int maf_atss_assign(const float* pd_bboxes, const float* anchor_points, const float* gts, const int32_t* gt_image, const int32_t* offsets,
                               int32_t T, int32_t B, int32_t A, int32_t topk, int32_t n_levels, const int32_t* level_hw, const float* level_stride,
                               float cell_offset, float cell_size, int32_t* cand_scratch, int32_t* out_gt, float* out_norm, maf_stream_t stream) {
    MAF_REQUIRE(pd_bboxes && anchor_points && offsets && out_gt && out_norm, "atss_assign: null pointer");
    MAF_REQUIRE(T == 0 || (gts && gt_image && cand_scratch), "atss_assign: null target pointer");
    MAF_REQUIRE(B > 0 && T >= 0 && A > 0 && A <= kMaxA && topk > 0 && topk <= 9 && n_levels <= 3, "atss_assign: bad shape (A <= 8400, topk <= 9, at most 3 levels)");
    TalArgs a = {};
    a.boxes = pd_bboxes; a.points = anchor_points; a.gts = gts; a.gt_img = gt_image; a.offs = offsets;
    a.cand = cand_scratch; a.out_gt = out_gt; a.out_norm = out_norm;
    a.A = A; a.nc = 1; a.B = B; a.alpha = 1.f; a.beta = 1.f; a.eps = 1e-9f;
    const int rc = set_levels(a, A, n_levels, level_hw, level_stride, cell_offset);
    if (rc) return rc;
    for (int l = 0; l < n_levels; ++l)                                          // the reference raises below topk anchors on a level (atss_assigner.py:104)
        MAF_REQUIRE(a.lh[l] >= 3 && a.lw[l] >= 3, "atss_assign: every level needs at least 3 x 3 anchors");
    a.atss = 1; a.half_cells = cell_size * 0.5f;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.topk = topk;
    if (T > 0) hipLaunchKernelGGL(atss_cand_kernel, dim3(T), dim3(kTK), 0, s, a);
    a.topk = topk * n_levels;                                                   // candidates per box for the resolve kernel
    hipLaunchKernelGGL(tal_resolve_kernel<float>, dim3(B), dim3(kTR), 0, s, a);
    return maf_check_hip(hipGetLastError(), "atss_assign launch");
}

extern "C" int maf_tal_assign(const void* pd_scores, int32_t score_dtype, const float* pd_bboxes, const float* anchor_points, const float* gts,
                              const int32_t* gt_image, const int32_t* offsets, int32_t T, int32_t B, int32_t A, int32_t nc, int32_t topk,
                              float alpha, float beta, float eps, int32_t n_levels, const int32_t* level_hw, const float* level_stride,
                              float cell_offset, int32_t* cand_scratch, int32_t* out_gt, float* out_norm, maf_stream_t stream) {
    MAF_REQUIRE(pd_scores && pd_bboxes && anchor_points && offsets && out_gt && out_norm, "tal_assign: null pointer");
    MAF_REQUIRE(T == 0 || (gts && gt_image && cand_scratch), "tal_assign: null target pointer");
    MAF_REQUIRE(score_dtype == MAF_F16 || score_dtype == MAF_F32, "tal_assign: scores must be f16 or f32");
    MAF_REQUIRE(B > 0 && T >= 0 && A > 0 && A <= kMaxA && nc > 0 && topk > 0 && topk <= A, "tal_assign: bad shape (at most 8400 anchors per image)");
    TalArgs a;
    a.scores = pd_scores; a.boxes = pd_bboxes; a.points = anchor_points; a.gts = gts; a.gt_img = gt_image; a.offs = offsets;
    a.cand = cand_scratch; a.out_gt = out_gt; a.out_norm = out_norm;
    a.A = A; a.nc = nc; a.topk = topk; a.B = B; a.alpha = alpha; a.beta = beta; a.eps = eps;
    MAF_REQUIRE(topk <= 32, "tal_assign: topk <= 32");
    const int rc = set_levels(a, A, n_levels, level_hw, level_stride, cell_offset);
    if (rc) return rc;
    a.atss = 0; a.half_cells = 0.f;
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

extern "C" int maf_tal_targets(const float* targets, int32_t T, int32_t B, float img_size, float* gts, int32_t* gt_image, int32_t* offsets,
                               maf_stream_t stream) {
    MAF_REQUIRE(offsets && (T == 0 || (targets && gts && gt_image)), "tal_targets: null pointer");
    MAF_REQUIRE(T >= 0 && T <= kMaxT && B > 0 && B <= kMaxB, "tal_targets: at most 16384 labels and 4096 images per batch");
    hipLaunchKernelGGL(tal_targets_kernel, dim3(1), dim3(kTR), 0, static_cast<hipStream_t>(stream), targets, T, B, img_size, gts, gt_image, offsets);
    return maf_check_hip(hipGetLastError(), "tal_targets launch");
}
