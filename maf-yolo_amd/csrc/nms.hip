// non_max_suppression for a whole batch in two launches, no host round trips.
//
// Replaces yolov6/utils/nms.py:31-105 (candidate filter :48, conf = obj*cls :69, xywh2xyxy :21-28,
// multi-label nonzero :75-77 / best-class :78-80, class filter :83-84, max_nms cap :90-91,
// class-offset trick :94-95, torchvision.ops.nms :96, max_det :97-98).  The reference runs a Python
// loop over images with ~6 implicit device->host syncs each.
//
// Launch 1  nms_collect:  one wavefront per box.  The 5+nc floats of a box are one contiguous,
//   coalesced read; the raw-class maximum / best class are wavefront reductions (DPP shuffles);
//   survivors are appended to the image's candidate list with one wave-aggregated atomic.  A
//   candidate is a 64-bit key  (~score_bits << 32) | (box*nc + cls): ascending key order ==
//   descending fp32 score, ties broken by the lower row-major (box, class) index — the order
//   `nonzero` (nms.py:76) + a stable descending sort produce.
// Launch 2  nms_select:   one workgroup per image.  Bitonic sort of the keys (in LDS up to 8192
//   keys, in global memory above that), top max_nms = 30000, then greedy NMS in score order with an
//   early exit at max_det kept boxes: 256 candidates at a time are screened against the kept list
//   (LDS) in parallel, then one wavefront resolves the chunk 64 candidates at a time with ballots —
//   "who is the next survivor" is a find-first-set on the wave's alive mask, so the serial part is
//   bounded by max_det + #chunks iterations, not by the candidate count.
//
// Arithmetic is kept bit-faithful to the reference's fp32 path: boxes are cx -/+ w/2 in fp32, the
// class offset is cls*4096 added in fp32 BEFORE the IoU, IoU = inter / (area_i + area_j - inter) with
// IEEE division and no FMA contraction (this file is compiled with -ffp-contract=off), and the
// quotient is compared with the threshold in double as torchvision's CPU kernel does.
#include "maf_common.h"

namespace {

constexpr int kMaxNms = 30000;       // nms.py:54
constexpr float kMaxWh = 4096.f;     // nms.py:53
constexpr int kLdsKeys = 8192;       // 64 KiB of 64-bit keys
constexpr int kMaxDetCap = 2048;     // kept-list capacity in LDS (5 floats each)

struct NmsArgs {
    const float* pred;
    int B, N, nc;
    float conf;
    double iou;
    const int* classes; int n_classes;
    int agnostic, multi_label, max_det;
    int* cnt;                 // [B]
    unsigned long long* keys; // [B][capP]
    long long capP;
    float* out_rows; long long* out_idx; int* out_count;
};

__device__ __forceinline__ bool class_ok(const NmsArgs& a, int c) {
    if (a.classes == nullptr) return true;
    for (int i = 0; i < a.n_classes; ++i)
        if (a.classes[i] == c) return true;
    return false;
}

__global__ __launch_bounds__(256) void nms_collect_kernel(const NmsArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int b = blockIdx.y;
    const int no = 5 + a.nc;
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    for (int box = wave; box < a.N; box += nwaves) {
        const float* row = a.pred + ((size_t)b * a.N + box) * no;
        const float obj = row[4];
        if (!(obj > a.conf)) continue;                           // nms.py:48 (wave-uniform)
        // raw class maximum over all classes (nms.py:48) and, for best-class mode, argmax of obj*cls
        float rmax = -INFINITY;
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int c = lane; c < a.nc; c += 64) {
            const float r = row[5 + c];
            rmax = fmaxf(rmax, r);
            const float sc = r * obj;                            // nms.py:69
            if (sc > best) { best = sc; besti = c; }             // first maximum per lane (c ascending)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rmax = fmaxf(rmax, __shfl_xor(rmax, o));
            const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(besti, o);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (!(rmax > a.conf)) continue;
        if (a.multi_label) {
            for (int c0 = 0; c0 < a.nc; c0 += 64) {
                const int c = c0 + lane;
                float sc = 0.f; bool ok = false;
                if (c < a.nc) {
                    sc = row[5 + c] * obj;
                    ok = sc > a.conf && class_ok(a, c);          // nms.py:76, :83-84
                }
                const unsigned long long m = __ballot(ok);
                if (m == 0) continue;
                int base = 0;
                if (lane == 0) base = atomicAdd(&a.cnt[b], __popcll(m));
                base = __shfl(base, 0);
                if (ok) {
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    const unsigned int flat = (unsigned int)box * a.nc + c;
                    keys[pos] = ((unsigned long long)(~__float_as_uint(sc)) << 32) | flat;
                }
            }
        } else if (lane == 0) {                                   // nms.py:78-80
            if (best > a.conf && class_ok(a, besti)) {
                const int pos = atomicAdd(&a.cnt[b], 1);
                const unsigned int flat = (unsigned int)box * a.nc + besti;
                keys[pos] = ((unsigned long long)(~__float_as_uint(best)) << 32) | flat;
            }
        }
    }
}

struct Cand { float x1, y1, x2, y2, area; };

__device__ __forceinline__ bool iou_gt(const Cand& k, const Cand& c, double thr) {
    // torchvision nms CPU kernel: i = kept (earlier), j = candidate
    const float xx1 = fmaxf(k.x1, c.x1), yy1 = fmaxf(k.y1, c.y1);
    const float xx2 = fminf(k.x2, c.x2), yy2 = fminf(k.y2, c.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (k.area + c.area - inter);
    return (double)ovr > thr;
}

__device__ void bitonic_sort(unsigned long long* k, int P, int tid, int nthreads) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i + j;
                const unsigned long long x = k[i], y = k[l];
                const bool up = (i & size) == 0;
                if ((x > y) == up) { k[i] = y; k[l] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void nms_select_kernel(const NmsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned long long lds_keys[kLdsKeys];   // reused as the kept list after the sort
    __shared__ unsigned long long alive_mask[4];
    __shared__ int s_kept;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const long long cap = a.multi_label ? (long long)a.N * a.nc : (long long)a.N;
    long long n64 = a.cnt[b];
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    float* rows = a.out_rows + (size_t)b * a.max_det * 6;
    long long* oidx = a.out_idx + (size_t)b * a.max_det;
    if (n == 0) {
        if (tid == 0) a.out_count[b] = 0;
        return;
    }
    int P = 1;
    while (P < n) P <<= 1;
    if (P <= kLdsKeys) {
        for (int i = tid; i < P; i += 256) lds_keys[i] = i < n ? keys[i] : ~0ull;
        __syncthreads();
        bitonic_sort(lds_keys, P, tid, 256);
        for (int i = tid; i < n; i += 256) keys[i] = lds_keys[i];
    } else {
        for (int i = n + tid; i < P; i += 256) keys[i] = ~0ull;
        __syncthreads();
        bitonic_sort(keys, P, tid, 256);
    }
    __syncthreads();

    const int ns = n < kMaxNms ? n : kMaxNms;                  // nms.py:90-91
    Cand* kept = reinterpret_cast<Cand*>(lds_keys);            // LDS reuse: [max_det] x 5 floats
    if (tid == 0) s_kept = 0;
    __syncthreads();
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;

    for (int base = 0; base < ns; base += 256) {
        const int kept0 = s_kept;
        if (kept0 >= a.max_det) break;
        // ---- phase A: all 256 lanes screen their candidate against the kept list ----
        const int ci = base + tid;
        Cand c = {0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned long long key = 0;
        bool alive = ci < ns;
        float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f, score = 0.f; int cls = 0; unsigned int flat = 0;
        if (alive) {
            key = keys[ci];
            flat = (unsigned int)(key & 0xffffffffu);
            score = __uint_as_float(~(unsigned int)(key >> 32));
            const unsigned int box = flat / (unsigned int)a.nc;
            cls = (int)(flat - box * (unsigned int)a.nc);
            const float* r = pred + (size_t)box * no;
            const float cx = r[0], cy = r[1], w = r[2], h = r[3];
            bx1 = cx - w / 2; by1 = cy - h / 2; bx2 = cx + w / 2; by2 = cy + h / 2;   // nms.py:21-28
            const float off = a.agnostic ? 0.f : (float)cls * kMaxWh;                 // nms.py:94
            c.x1 = bx1 + off; c.y1 = by1 + off; c.x2 = bx2 + off; c.y2 = by2 + off;
            c.area = (c.x2 - c.x1) * (c.y2 - c.y1);
            for (int k = 0; k < kept0 && alive; ++k)
                if (iou_gt(kept[k], c, a.iou)) alive = false;
        }
        const unsigned long long m = __ballot(alive);
        if (lane == 0) alive_mask[wave] = m;
        __syncthreads();
        // ---- phase B: resolve the chunk in score order, one wavefront of 64 candidates at a time.
        // Every wave runs its own sub-chunk in turn so each lane keeps its candidate in registers.
        for (int sub = 0; sub < 4; ++sub) {
            if (wave == sub) {
                int nk = s_kept;
                // survivors kept by earlier sub-chunks of this chunk
                for (int k = kept0; k < nk && alive; ++k)
                    if (iou_gt(kept[k], c, a.iou)) alive = false;
                unsigned long long mask = __ballot(alive);
                while (mask != 0 && nk < a.max_det) {
                    const int i = __ffsll((long long)mask) - 1;           // next survivor in score order
                    Cand ki;
                    ki.x1 = __shfl(c.x1, i); ki.y1 = __shfl(c.y1, i); ki.x2 = __shfl(c.x2, i); ki.y2 = __shfl(c.y2, i);
                    ki.area = __shfl(c.area, i);
                    if (lane == i) {
                        kept[nk] = c;
                        float* o = rows + (size_t)nk * 6;
                        o[0] = bx1; o[1] = by1; o[2] = bx2; o[3] = by2; o[4] = score; o[5] = (float)cls;
                        oidx[nk] = a.multi_label ? (long long)flat : (long long)(flat / (unsigned int)a.nc);
                        alive = false;
                    }
                    ++nk;
                    if (alive && lane > i && iou_gt(ki, c, a.iou)) alive = false;
                    mask = __ballot(alive) & ~((2ull << i) - 1ull);       // only later candidates remain
                    // (earlier lanes are already kept or suppressed)
                }
                if (lane == 0) s_kept = nk;
            }
            __syncthreads();
        }
    }
    if (tid == 0) a.out_count[b] = s_kept < a.max_det ? s_kept : a.max_det;
}

long long pow2ceil(long long v) {
    long long p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" int64_t maf_nms_workspace_bytes(int32_t B, int32_t N, int32_t nc) {
    if (B <= 0 || N <= 0 || nc <= 0) return 0;
    const long long capP = pow2ceil((long long)N * nc);
    return 256 + (long long)((B * 4 + 255) / 256) * 256 + (long long)B * capP * 8;
}

extern "C" int maf_nms(const float* pred, int32_t B, int32_t N, int32_t nc, double conf_thres, double iou_thres,
                       const int32_t* classes, int32_t n_classes, int32_t agnostic, int32_t multi_label,
                       int32_t max_det, void* workspace, int64_t workspace_bytes,
                       float* out_rows, int64_t* out_idx, int32_t* out_count, maf_stream_t stream) {
    MAF_REQUIRE(pred && workspace && out_rows && out_idx && out_count, "nms: null pointer");
    MAF_REQUIRE(B > 0 && N > 0 && nc > 0, "nms: bad shape");
    MAF_REQUIRE((long long)N * nc < (1ll << 32), "nms: N*nc must fit 32 bits");
    MAF_REQUIRE(conf_thres >= 0.0 && conf_thres <= 1.0, "nms: conf_thres must be in [0,1] (nms.py:50)");
    MAF_REQUIRE(iou_thres >= 0.0 && iou_thres <= 1.0, "nms: iou_thres must be in [0,1] (nms.py:51)");
    MAF_REQUIRE(max_det > 0 && max_det <= kMaxDetCap, "nms: max_det must be in 1..2048");
    MAF_REQUIRE(workspace_bytes >= maf_nms_workspace_bytes(B, N, nc), "nms: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    NmsArgs a;
    a.pred = pred; a.B = B; a.N = N; a.nc = nc;
    a.conf = (float)conf_thres;            // torch compares the fp32 tensor against the scalar in fp32
    a.iou = iou_thres;
    a.classes = n_classes > 0 ? classes : nullptr; a.n_classes = n_classes;
    a.agnostic = agnostic; a.multi_label = (multi_label && nc > 1) ? 1 : 0;    // nms.py:57
    a.max_det = max_det;
    char* ws = static_cast<char*>(workspace);
    a.cnt = reinterpret_cast<int*>(ws);
    a.keys = reinterpret_cast<unsigned long long*>(ws + 256 + (long long)((B * 4 + 255) / 256) * 256);
    a.capP = pow2ceil((long long)N * nc);
    a.out_rows = out_rows; a.out_idx = reinterpret_cast<long long*>(out_idx); a.out_count = out_count;
    int rc = maf_check_hip(hipMemsetAsync(a.cnt, 0, (size_t)B * 4, s), "nms memset");
    if (rc) return rc;
    const int blocks_x = (N + 3) / 4 < 2048 ? (N + 3) / 4 : 2048;    // 4 waves per block, one box per wave-iteration
    hipLaunchKernelGGL(nms_collect_kernel, dim3(blocks_x, B), dim3(256), 0, s, a);
    rc = maf_check_hip(hipGetLastError(), "nms_collect launch");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_select_kernel, dim3(B), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "nms_select launch");
}
