// non_max_suppression for a whole batch in two launches, no host round trips.
//
// Replaces yolov6/utils/nms.py:31-105 (candidate filter :48, conf = obj*cls :69, xywh2xyxy :21-28,
// multi-label nonzero :75-77 / best-class :78-80, class filter :83-84, max_nms cap :90-91,
// class-offset trick :94-95, torchvision.ops.nms :96, max_det :97-98).  The reference runs a Python
// loop over images with ~6 implicit device->host syncs each.
//
// Launch 1  nms_collect:  multi-label mode streams the prediction tensor one lane per (box, class)
//   element, fully coalesced; best-class mode gives each box a 16-lane group and finds the arg-max
//   with shuffles.  Survivors are appended to the image's candidate list with one wave-aggregated
//   atomic per wavefront.  A
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
constexpr int kMaxDetCap = 1024;     // kept-list capacity in LDS (32 B each = lower half of the sort buffer)
constexpr int kMaxBands = 1022;      // class bands of the cls*4096 offset trick handled by the band index

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

// raw-class maximum of one box (only needed when obj > 1, where sc > conf does not imply it)
__device__ float row_rmax(const float* row, int nc) {
    float m = -INFINITY;
    for (int c = 0; c < nc; ++c) m = fmaxf(m, row[5 + c]);
    return m;
}

constexpr int kCntStride = 64;       // one candidate counter per 256-byte line: atomics of different images never share a line
constexpr int kChunk = 8192;         // elements per workgroup chunk (32 KiB of fp32: stays in L1 between the two passes)

__device__ __forceinline__ bool multi_test(const NmsArgs& a, const float* pred, unsigned int e32, int no, float& sc) {
    const int box = (int)(e32 / (unsigned int)a.nc), c = (int)(e32 - (unsigned int)box * (unsigned int)a.nc);
    const float* row = pred + (size_t)box * no;
    const float obj = row[4];
    sc = row[5 + c] * obj;                                                // nms.py:69
    bool ok = obj > a.conf && sc > a.conf && class_ok(a, c);              // nms.py:48 (obj), :76, :83-84
    // nms.py:48 also needs max(raw cls) > conf; sc > conf implies it unless obj > 1
    if (ok && obj > 1.0f) ok = row_rmax(row, a.nc) > a.conf;
    return ok;
}

// multi_label: one lane per (box, class) element of the prediction tensor, fully coalesced.  A workgroup owns a
// contiguous chunk: pass 1 counts its candidates, ONE atomicAdd per chunk reserves the output range (a per-wave
// atomic per hit costs ~10 ns each on one contended line — 64k of them were 0.65 ms), pass 2 re-tests (L1 hits)
// and writes the keys.  Candidate order inside the list is irrelevant: the sort key carries the flat index.
__global__ __launch_bounds__(256) void nms_collect_multi_kernel(const NmsArgs a) {
    __shared__ int wave_cnt[4];
    __shared__ int s_base;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int no = 5 + a.nc;
    const unsigned int total = (unsigned int)a.N * (unsigned int)a.nc;     // < 2^32 (checked by maf_nms)
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const float* pred = a.pred + (size_t)b * a.N * no;
    for (unsigned int c0 = blockIdx.x * kChunk; c0 < total; c0 += gridDim.x * kChunk) {
        int mine = 0;
        for (int i = 0; i < kChunk / 256; ++i) {
            const unsigned int e = c0 + i * 256 + tid;
            float sc;
            const bool ok = e < total && multi_test(a, pred, e, no, sc);
            mine += __popcll(__ballot(ok));                                // wave-uniform running count
        }
        if (lane == 0) wave_cnt[wave] = mine;
        __syncthreads();
        if (tid == 0) {
            const int n = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            s_base = n > 0 ? atomicAdd(&a.cnt[b * kCntStride], n) : 0;
        }
        __syncthreads();
        int pos = s_base;
        for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
        const bool any = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3] > 0;
        if (any) {
            for (int i = 0; i < kChunk / 256; ++i) {
                const unsigned int e = c0 + i * 256 + tid;
                float sc = 0.f;
                const bool ok = e < total && multi_test(a, pred, e, no, sc);
                const unsigned long long m = __ballot(ok);
                if (ok) keys[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(~__float_as_uint(sc)) << 32) | e;
                pos += __popcll(m);
            }
        }
        __syncthreads();
    }
}

// best-class mode: 16 lanes per box (4 boxes per wavefront), arg-max by shuffles within the 16-lane group.
__global__ __launch_bounds__(256) void nms_collect_best_kernel(const NmsArgs a) {
    const int b = blockIdx.y;
    const int sub = threadIdx.x & 15;
    const int no = 5 + a.nc;
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const int groups = (gridDim.x * blockDim.x) >> 4;
    for (int box = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; box < a.N; box += groups) {
        const float* row = a.pred + ((size_t)b * a.N + box) * no;
        const float obj = row[4];
        float rmax = -INFINITY, best = -INFINITY; int besti = 0x7fffffff;
        for (int c = sub; c < a.nc; c += 16) {
            const float r = row[5 + c];
            rmax = fmaxf(rmax, r);
            const float sc = r * obj;                                     // nms.py:69
            if (sc > best) { best = sc; besti = c; }                      // first maximum per lane (c ascending)
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            rmax = fmaxf(rmax, __shfl_xor(rmax, o));
            const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(besti, o);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (sub == 0 && obj > a.conf && rmax > a.conf && best > a.conf && class_ok(a, besti)) {   // nms.py:48, :78-80, :83-84
            const int pos = atomicAdd(&a.cnt[b * kCntStride], 1);
            const unsigned int flat = (unsigned int)box * a.nc + besti;
            keys[pos] = ((unsigned long long)(~__float_as_uint(best)) << 32) | flat;
        }
    }
}

struct alignas(16) Cand { float x1, y1, x2, y2, area, score; unsigned int flat, pad; };   // 32 B: two ds_read_b128

__device__ __forceinline__ bool iou_gt(const Cand& k, const Cand& c, double thr) {
    // torchvision nms CPU kernel: i = kept (earlier), j = candidate
    const float xx1 = fmaxf(k.x1, c.x1), yy1 = fmaxf(k.y1, c.y1);
    const float xx2 = fminf(k.x2, c.x2), yy2 = fminf(k.y2, c.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    // disjoint boxes: inter = 0 (or NaN) => ovr is 0, -0 or NaN => never > thr (thr >= 0). Skips the IEEE division
    // for the vast majority of pairs (different classes sit 4096 px apart).
    if (!(w > 0.f) || !(h > 0.f)) return false;
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

// Band index over the kept list.  With the class-offset trick (nms.py:94-95) a box of class c lives around
// x in [c*4096, (c+1)*4096): two boxes can only intersect if their x-intervals share a 4096-px band.  Every kept box
// is linked into the list(s) of the band(s) its x-interval touches (or into one global list if it spans more than
// two), a candidate walks only the lists of its own bands + the global list.  Exact for any input (bands are
// clamped the same way on both sides); turns the candidate-vs-kept screening from O(kept) into O(kept of my class).
struct BandIndex {
    short* head;       // [nb + 1]  (last = global list)
    short* next;       // [2 * kMaxDetCap]
    short* box;        // [2 * kMaxDetCap]
    int nb;
};

__device__ __forceinline__ int band_of(float v, int nb) {
    const float f = floorf(v * (1.0f / kMaxWh)) + 1.0f;          // bands -1 .. nc  ->  0 .. nb-1
    return !(f > 0.f) ? 0 : (f >= (float)(nb - 1) ? nb - 1 : (int)f);
}

__device__ __forceinline__ void band_range(const Cand& c, int nb, int& lo, int& hi) {
    lo = band_of(fminf(c.x1, c.x2), nb);
    hi = band_of(fmaxf(c.x1, c.x2), nb);
}

__global__ __launch_bounds__(256) void nms_select_kernel(const NmsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned long long lds_keys[kLdsKeys];   // reused as the kept list after the sort
    __shared__ unsigned long long alive_mask[4];
    __shared__ int s_kept;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const long long cap = a.multi_label ? (long long)a.N * a.nc : (long long)a.N;
    long long n64 = a.cnt[b * kCntStride];
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
    Cand* kept = reinterpret_cast<Cand*>(lds_keys);            // LDS reuse: lower 32 KiB = kept list [<= 1024] x 32 B
    BandIndex bi;                                              //            upper 32 KiB = band index
    bi.nb = (a.agnostic || a.nc + 2 > kMaxBands) ? 1 : a.nc + 2;
    bi.head = reinterpret_cast<short*>(lds_keys + kLdsKeys / 2);
    bi.next = bi.head + 1024;
    bi.box = bi.next + 2 * kMaxDetCap;
    for (int i = tid; i <= bi.nb; i += 256) bi.head[i] = -1;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;

    for (int base = 0; base < ns; base += 256) {
        const int kept0 = s_kept;
        if (kept0 >= a.max_det) break;
        // ---- phase A: all 256 lanes screen their candidate against the kept list ----
        const int ci = base + tid;
        Cand c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
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
            c.score = score; c.flat = flat;
            int lo, hi;
            band_range(c, bi.nb, lo, hi);
            if (hi - lo > 1) {                                             // pathological span: scan the whole list
                for (int k = 0; k < kept0 && alive; ++k)
                    if (iou_gt(kept[k], c, a.iou)) alive = false;
            } else {
                for (int pass = 0; pass < 3 && alive; ++pass) {            // my band(s), then the global list
                    const int bnd = pass == 0 ? lo : pass == 1 ? hi : bi.nb;
                    if (pass == 1 && hi == lo) continue;
                    int guard = 2 * kMaxDetCap;                            // a list can never be longer than the node pool
                    for (int nd = bi.head[bnd]; nd >= 0 && alive && guard-- > 0; nd = bi.next[nd])
                        if (iou_gt(kept[bi.box[nd]], c, a.iou)) alive = false;
                }
            }
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
                        kept[nk] = c;                                     // rows are written after the loop, from the list
                        int lo, hi;
                        band_range(c, bi.nb, lo, hi);
                        // Lanes take turns here and hand the list heads to each other through LDS inside one wave:
                        // that is a cross-thread hand-off without a barrier, so the accesses must be volatile (the
                        // compiler may otherwise keep a lane-private copy; a stale node counter made a self-loop).
                        // Node ids are derived from nk (2 per kept box), not from a shared counter.
                        volatile short* vhead = bi.head;
                        if (hi - lo > 1) { lo = hi = bi.nb; }            // spans > 2 bands: global list
                        const int n0 = 2 * nk;
                        bi.box[n0] = (short)nk; bi.next[n0] = vhead[lo]; vhead[lo] = (short)n0;
                        if (hi != lo) { bi.box[n0 + 1] = (short)nk; bi.next[n0 + 1] = vhead[hi]; vhead[hi] = (short)(n0 + 1); }
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
    // ---- emit rows (x1,y1,x2,y2,conf,cls) of the survivors, un-offset boxes recomputed from the prediction (nms.py:21-28)
    const int nk = s_kept < a.max_det ? s_kept : a.max_det;
    for (int k = tid; k < nk; k += 256) {
        const unsigned int flat = kept[k].flat;
        const unsigned int box = flat / (unsigned int)a.nc;
        const int cls = (int)(flat - box * (unsigned int)a.nc);
        const float* r = pred + (size_t)box * no;
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float* o = rows + (size_t)k * 6;
        o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2; o[4] = kept[k].score; o[5] = (float)cls;
        oidx[k] = a.multi_label ? (long long)flat : (long long)box;
    }
    if (tid == 0) a.out_count[b] = nk;
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
    return 256 + (long long)B * kCntStride * 4 + (long long)B * capP * 8;
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
    MAF_REQUIRE(max_det > 0 && max_det <= kMaxDetCap, "nms: max_det must be in 1..1024");
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
    a.keys = reinterpret_cast<unsigned long long*>(ws + 256 + (long long)B * kCntStride * 4);
    a.capP = pow2ceil((long long)N * nc);
    a.out_rows = out_rows; a.out_idx = reinterpret_cast<long long*>(out_idx); a.out_count = out_count;
    int rc = maf_check_hip(hipMemsetAsync(a.cnt, 0, (size_t)B * kCntStride * 4, s), "nms memset");
    if (rc) return rc;
    if (a.multi_label) {
        const long long el = (long long)N * nc;
        const int bx = (int)((el + kChunk - 1) / kChunk < 256 ? (el + kChunk - 1) / kChunk : 256);
        hipLaunchKernelGGL(nms_collect_multi_kernel, dim3(bx, B), dim3(256), 0, s, a);
    } else {
        const int bx = (N + 15) / 16 < 1024 ? (N + 15) / 16 : 1024;      // 16 boxes per 256-thread block per iteration
        hipLaunchKernelGGL(nms_collect_best_kernel, dim3(bx, B), dim3(256), 0, s, a);
    }
    rc = maf_check_hip(hipGetLastError(), "nms_collect launch");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_select_kernel, dim3(B), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "nms_select launch");
}
