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
// Launch 2  nms_select:   one workgroup per image; see the kernel comment (per-class fast path, generic path).
//
// Arithmetic is kept bit-faithful to the reference's fp32 path: boxes are cx -/+ w/2 in fp32, the
// class offset is cls*4096 added in fp32 BEFORE the IoU, IoU = inter / (area_i + area_j - inter) with
// IEEE division and no FMA contraction (this file is compiled with -ffp-contract=off), and the
// quotient is compared with the threshold in double as torchvision's CPU kernel does.
#include "maf_common.h"
#include <cmath>
#include <cstring>

namespace {

constexpr int kMaxNms = 30000;       // nms.py:54
constexpr float kMaxWh = 4096.f;     // nms.py:53
constexpr int kLdsKeys = 8192;       // 64 KiB of 64-bit keys
constexpr int kMaxDetCap = 1024;     // kept-list capacity in LDS (32 B each = lower half of the sort buffer)
constexpr int kMaskN = 4096;         // images with at most this many candidates take the suppression-matrix path
constexpr int kMaskW = kMaskN / 64;  // 64-bit words per matrix row
constexpr int kMaskWgs = 1024;       // matrix workgroups (one wave each) per image: ~2000 candidates are 528 block pairs — with 256 workgroups 16 of them
                                     // took a third pair and set the kernel's time (NMS of a 32-image batch 0.256 -> 0.224 ms with one pair per wave)

struct NmsArgs {
    const float* pred;
    int B, N, nc;
    float conf;
    double iou;
    double iou_m; int iou_even;   // division-free exact form of `fl(inter/union) > iou` (see iou_gt)
    const int* classes; int n_classes;
    int agnostic, multi_label, max_det;
    int* cnt;                 // [B]
    unsigned long long* keys; // [B][capP]
    long long capP;
    float* out_rows; long long* out_idx; int* out_count;
    struct Cand* cands;            // [B][kMaskN]           sorted candidates of the matrix path
    unsigned long long* mask;      // [B][kMaskN][kMaskW]   bit j of word w of row i: candidate 64w+j (> i) overlaps candidate i
    int cpath_ok, bbits;           // per-class path allowed (not agnostic, key fields fit); bits of a box index
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

constexpr int kCntStride = MAF_NMS_CNT_STRIDE;   // (64) one candidate counter per 256-byte line: atomics of different images never share a line
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

// the same test on values already loaded (obj = row[4], raw = row[5 + c])
__device__ __forceinline__ bool multi_ok(const NmsArgs& a, const float* pred, unsigned int e32, int no, float obj, float raw) {
    const float sc = raw * obj;                                           // nms.py:69
    const int box = (int)(e32 / (unsigned int)a.nc), c = (int)(e32 - (unsigned int)box * (unsigned int)a.nc);
    bool ok = obj > a.conf && sc > a.conf && class_ok(a, c);              // nms.py:48 (obj), :76, :83-84
    if (ok && obj > 1.0f) ok = row_rmax(pred + (size_t)box * no, a.nc) > a.conf;
    return ok;
}

// multi_label: one lane per (box, class) element of the prediction tensor, fully coalesced.  A workgroup owns a contiguous chunk of
// 8192 elements (32 per lane): it reads them once, keeps the 32 scores and a hit mask in registers, counts the candidates with wave
// ballots, reserves the output range with ONE atomicAdd per chunk (a per-wave atomic per hit costs ~10 ns each on one contended line —
// 64k of them were 0.65 ms) and writes the keys from the registers.  Candidate order inside the list is irrelevant: the sort key carries
// the flat index.
__device__ __forceinline__ void nms_collect_multi_body(const NmsArgs& a) {
    __shared__ int wave_cnt[4];
    __shared__ int s_base;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int no = 5 + a.nc;
    const unsigned int total = (unsigned int)a.N * (unsigned int)a.nc;     // < 2^32 (checked by maf_nms)
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const float* pred = a.pred + (size_t)b * a.N * no;
    for (unsigned int c0 = blockIdx.x * kChunk; c0 < total; c0 += gridDim.x * kChunk) {
        // ONE pass over memory: 8 elements per lane are loaded together (unconditional, clamped), tested, and their scores stay in
        // registers with a hit mask; after the chunk's single atomicAdd the keys are written from the registers.
        constexpr int E = kChunk / 256;                                    // 32 elements per lane
        float sc[E];
        unsigned int hits = 0;
        int mine = 0;
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += 8) {
            float obj[8], raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned int e = min(c0 + (i0 + u) * 256 + tid, total - 1);
                const unsigned int box = e / (unsigned int)a.nc;
                const float* row = pred + (size_t)box * no;
                obj[u] = row[4]; raw[u] = row[5 + (e - box * (unsigned int)a.nc)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned int e = c0 + (i0 + u) * 256 + tid;
                const bool ok = e < total && multi_ok(a, pred, e, no, obj[u], raw[u]);
                sc[i0 + u] = raw[u] * obj[u];
                hits |= ok ? 1u << (i0 + u) : 0u;
                mine += __popcll(__ballot(ok));                            // wave-uniform running count
            }
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
        if (mine > 0) {                                                    // wave-uniform
#pragma unroll
            for (int u = 0; u < E; ++u) {
                const bool ok = (hits >> u) & 1u;
                const unsigned long long m = __ballot(ok);
                if (ok) keys[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(~__float_as_uint(sc[u])) << 32) | (c0 + u * 256 + tid);
                pos += __popcll(m);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void nms_collect_multi_kernel(const NmsArgs a) { nms_collect_multi_body(a); }

// best-class mode: 16 lanes per box (4 boxes per wavefront), arg-max by shuffles within the 16-lane group.
__device__ __forceinline__ void nms_collect_best_body(const NmsArgs& a) {
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

__global__ __launch_bounds__(256) void nms_collect_best_kernel(const NmsArgs a) { nms_collect_best_body(a); }

struct alignas(16) Cand { float x1, y1, x2, y2, area, score; unsigned int flat, pad; };   // 32 B: two ds_read_b128

// Exact `fl32(inter / uni) > thr` without the IEEE division (~40 dependent instructions, the dominant cost of the
// same-class tests).  Let tf be the smallest fp32 above thr, p its predecessor, m = (tf + p) / 2.  The rounded quotient
// is >= tf  <=>  inter/uni > m, or == m with tf the even neighbour.  m has <= 25 significant bits and uni 24, so
// m * uni is exact in double: the comparison below involves no rounding at all.
struct IouThr { double thr, m; int even; float m32; };   // m32 = fl32(m): the single-precision screen below

__device__ __forceinline__ bool iou_gt(const Cand& k, const Cand& c, const IouThr& t) {
    // torchvision nms CPU kernel: i = kept (earlier), j = candidate
    const float xx1 = fmaxf(k.x1, c.x1), yy1 = fmaxf(k.y1, c.y1);
    const float xx2 = fminf(k.x2, c.x2), yy2 = fminf(k.y2, c.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    // disjoint boxes: inter = 0 (or NaN) => ovr is 0, -0 or NaN => never > thr (thr >= 0)
    if (!(w > 0.f) || !(h > 0.f)) return false;
    const float inter = w * h;
    const float uni = k.area + c.area - inter;
    // single-precision screen: d = inter - fl(m) * uni (one rounding) differs from inter - m * uni by at most 6e-8 * (uni + |d|), so outside
    // a band of 3e-7 * uni its sign is the exact answer; inside the band (and for NaN / infinite / non-positive unions, which never pass
    // this comparison) the exact double form below decides.  Most same-class pairs are far outside the band: no double arithmetic.
    const float d = __builtin_fmaf(-t.m32, uni, inter);
    if (uni > 0.f && fabsf(d) > 3e-7f * uni) return d > 0.f;
    if (uni > 0.f && uni < INFINITY && inter < INFINITY) {
        const double pm = t.m * (double)uni, di = (double)inter;
        return di > pm || (di == pm && t.even);
    }
    return (double)(inter / uni) > t.thr;                   // zero / negative / non-finite union: the literal formula
}

// The screen alone, branch-free, for the pair-matrix kernel: 1 / 0 = decided, -1 = inside the band (or a NaN / infinite / non-positive
// union: every comparison below is false for those) — the caller then asks iou_gt.  Disjoint boxes give inter = 0, d = -m * uni < 0: "no".
__device__ __forceinline__ int iou_screen(const Cand& k, const Cand& c, float m32) {
    const float w = fmaxf(0.f, fminf(k.x2, c.x2) - fmaxf(k.x1, c.x1)), h = fmaxf(0.f, fminf(k.y2, c.y2) - fmaxf(k.y1, c.y1));
    const float inter = w * h, uni = k.area + c.area - inter;
    const float d = __builtin_fmaf(-m32, uni, inter);
    return (uni > 0.f && fabsf(d) > 3e-7f * uni) ? (d > 0.f ? 1 : 0) : -1;
}

// Bitonic sort of P (a power of two) 64-bit keys by `nthreads` (a multiple of 64) threads of one workgroup.  Pair t of a pass with distance j is
// (i, i + j), i = ((t & ~(j - 1)) << 1) | (t & (j - 1)): for j <= 64 the 64 consecutive pairs a wave takes lie inside ONE aligned window of 128 keys, the same window in
// every such pass — so between two passes that both have j <= 64 a wave only has to see its OWN stores: a wavefront fence instead of a workgroup barrier (LDS: `lds`
// = true; a wave's LDS operations execute in order).  Of the 66 passes of a 2 048-key sort 51 are wave-local: 21 barriers instead of 66 (round 6; measured: NMS of a batch
// -3 us — the sort is bound by its LDS round trips per pass, not by the barriers).
// Keys in global memory (`lds` = false: the > 8 192-candidate form) keep a barrier after every pass.
__device__ void bitonic_sort(unsigned long long* k, int P, int tid, int nthreads, bool lds = false) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i + j;
                const unsigned long long x = k[i], y = k[l];
                const bool up = (i & size) == 0;
                if ((x > y) == up) { k[i] = y; k[l] = x; }
            }
            const int jn = j > 1 ? j >> 1 : size;                    // distance of the pass that follows (the next size opens with j = size; past the last pass: a barrier)
            if (lds && j <= 64 && jn <= 64 && (j > 1 || (size << 1) <= P)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                __syncthreads();
            }
        }
    }
}

__device__ __forceinline__ void make_cand(const NmsArgs& a, const float* pred, int no, unsigned int box, int cls, Cand& c) {
    const float* r = pred + (size_t)box * no;
    const float cx = r[0], cy = r[1], w = r[2], h = r[3];
    const float bx1 = cx - w / 2, by1 = cy - h / 2, bx2 = cx + w / 2, by2 = cy + h / 2;   // nms.py:21-28
    const float off = a.agnostic ? 0.f : (float)cls * kMaxWh;                             // nms.py:94
    c.x1 = bx1 + off; c.y1 = by1 + off; c.x2 = bx2 + off; c.y2 = by2 + off;
    c.area = (c.x2 - c.x1) * (c.y2 - c.y1);
}

constexpr int kSelT = 256;
__device__ unsigned long long g_nms_dbg[8];   // phase cycle stamps of image 0 (diagnostics: maf_nms_debug)

// One workgroup per image.
//   1. bitonic sort of the 64-bit keys (LDS up to 8192 keys, global memory above), top max_nms = 30000;
//   2. greedy NMS in score order with an early exit at max_det survivors.  Survivors live in LDS (`kept`) and are linked
//      into per-class lists (`head`/`next`, newest first).  256 candidates at a time walk the list of their class in
//      parallel (phase A); then one wavefront at a time resolves its 64 candidates (phase B):
//        - walk only the list entries created since phase A,
//        - lanes of equal class find each other with a ballot-built match mask, each lane tests the IoU against the
//          earlier alive lanes of its class and records them as a 64-bit "suppressed-by" set,
//        - a scalar scan in lane order decides survivors: lane j survives iff none of its suppressors survived,
//        - survivors append themselves to `kept` and to their class list in parallel.
// Per-class lists are exact only if boxes of different classes cannot intersect.  With the class-offset trick
// (nms.py:94-95) a box of class c sits at x + c*4096, so that holds whenever the x-extent of ALL candidates of the image
// is <= 4095 px (also under the fp32 rounding of the offset add, which is monotone).  One reduction per image decides;
// otherwise (or agnostic, or nc > 1024) every box goes to ONE list and every pair is tested, as torchvision does.
__device__ __forceinline__ void nms_select_body(const NmsArgs& a, const int b) {
    __shared__ __attribute__((aligned(16))) unsigned long long lds_keys[kLdsKeys];   // sort buffer, then: kept list (lower half)
    // after the sort the upper half of the buffer holds: the chunk's candidates (cross-lane reads), list heads, list links
    Cand* wbox = reinterpret_cast<Cand*>(lds_keys + kLdsKeys / 2);                    // [256] x 32 B
    int* head = reinterpret_cast<int*>(wbox + kSelT);                                 // [1024]
    short* nxt = reinterpret_cast<short*>(head + 1024);                               // [kMaxDetCap]
    int* cstart = reinterpret_cast<int*>(nxt + kMaxDetCap);                           // [1025] class-sorted view of the kept list:
    int* cfill = cstart + 1028;                                                       // [1024]   start offsets / fill cursors
    short* order = reinterpret_cast<short*>(cfill + 1024);                            // [kMaxDetCap] kept slots grouped by class
    __shared__ int s_kept, s_wide;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const long long cap = (long long)a.N * a.nc;
    long long n64 = a.cnt[b * kCntStride];
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    float* rows = a.out_rows + (size_t)b * a.max_det * 6;
    long long* oidx = a.out_idx + (size_t)b * a.max_det;
    if (n == 0) {
        if (tid == 0) a.out_count[b] = 0;
        return;
    }
    if (n <= kMaskN && a.mask) return;                         // this image takes the suppression-matrix path (kernels below)
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;
    int P = 1;
    while (P < n) P <<= 1;
    const IouThr iouthr = {a.iou, a.iou_m, a.iou_even, (float)a.iou_m};
    unsigned long long t_a = 0, t_b = 0, t_ld = 0, t0 = __builtin_readcyclecounter();

    // ---- can classes interact in this image?  (x-extent of all candidates)
    if (tid == 0) s_wide = (a.agnostic || a.nc > 1024) ? 1 : 0;
    __syncthreads();
    if (!a.agnostic && a.nc <= 1024) {
        float lo = INFINITY, hi = -INFINITY;
        bool bad = false;
        for (int i = tid; i < n; i += kSelT) {
            const unsigned int flat = (unsigned int)(keys[i] & 0xffffffffu);
            const float* r = pred + (size_t)(flat / (unsigned int)a.nc) * no;
            const float x1 = r[0] - r[2] / 2, x2 = r[0] + r[2] / 2;
            if (!(x1 == x1) || !(x2 == x2)) bad = true;
            lo = fminf(lo, fminf(x1, x2)); hi = fmaxf(hi, fmaxf(x1, x2));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        if (lane == 0 && (bad || !((double)hi - (double)lo <= 4095.0))) s_wide = 1;
    }

    if (P <= kLdsKeys) {
        for (int i = tid; i < P; i += kSelT) lds_keys[i] = i < n ? keys[i] : ~0ull;
        __syncthreads();
        bitonic_sort(lds_keys, P, tid, kSelT, true);
        for (int i = tid; i < n; i += kSelT) keys[i] = lds_keys[i];
    } else {
        for (int i = n + tid; i < P; i += kSelT) keys[i] = ~0ull;
        __syncthreads();
        bitonic_sort(keys, P, tid, kSelT);
    }
    __syncthreads();
    const bool by_class = s_wide == 0;                         // uniform
    const unsigned long long t_sorted = __builtin_readcyclecounter();

    const int ns = n < kMaxNms ? n : kMaxNms;                  // nms.py:90-91
    Cand* kept = reinterpret_cast<Cand*>(lds_keys);            // LDS reuse: kept list [<= 1024] x 32 B
    if (tid == 0) s_kept = 0;
    for (int i = tid; i < 1024; i += kSelT) head[i] = -1;
    __syncthreads();

    for (int base = 0; base < ns; base += kSelT) {
        const int kept0 = s_kept;
        if (kept0 >= a.max_det) break;
        const unsigned long long ta0 = __builtin_readcyclecounter();
        // ---- group the survivors so far by class (counting sort: contiguous ranges instead of pointer chasing) ----
        const int nlists = by_class ? a.nc : 1;
        for (int i = tid; i <= nlists; i += kSelT) { cstart[i] = 0; if (i < nlists) cfill[i] = 0; }
        __syncthreads();
        for (int k = tid; k < kept0; k += kSelT) atomicAdd(&cstart[kept[k].pad + 1], 1);
        __syncthreads();
        if (wave == 0) {                                       // inclusive scan of the (<= 1024) class counts by one wavefront
            int carry = 0;
            for (int i0 = 0; i0 < nlists; i0 += 64) {
                const int i = i0 + lane;
                int v = i < nlists ? cstart[i + 1] : 0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
                if (i < nlists) cstart[i + 1] = v + carry;
                carry += __shfl(v, 63);
            }
        }
        __syncthreads();
        for (int k = tid; k < kept0; k += kSelT) {
            const int cl = (int)kept[k].pad;
            order[cstart[cl] + atomicAdd(&cfill[cl], 1)] = (short)k;
        }
        __syncthreads();
        // ---- phase A: every lane scans the survivors of its class ----
        const int ci = base + tid;
        Cand c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
        bool alive = ci < ns;
        int lst = 0;                                           // list this candidate belongs to
        if (alive) {
            const unsigned long long key = keys[ci];
            const unsigned int flat = (unsigned int)(key & 0xffffffffu);
            const unsigned int box = flat / (unsigned int)a.nc;
            const int cls = (int)(flat - box * (unsigned int)a.nc);
            make_cand(a, pred, no, box, cls, c);
            c.score = __uint_as_float(~(unsigned int)(key >> 32)); c.flat = flat;
            lst = by_class ? cls : 0;
            c.pad = (unsigned int)lst;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_ld += __builtin_readcyclecounter() - ta0;
            const int p1 = cstart[lst + 1];
            for (int p = cstart[lst]; p < p1 && alive; p += 4) {           // 4 independent LDS gathers per step
                bool hit = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) hit |= iou_gt(kept[order[min(p + u, p1 - 1)]], c, iouthr);
                if (hit) alive = false;
            }
        }
        wbox[tid] = c;
        __syncthreads();
        const unsigned long long tb0 = __builtin_readcyclecounter();
        t_a += tb0 - ta0;
        // ---- phase B: one wavefront of 64 candidates at a time, in score order
        for (int sub = 0; sub < kSelT / 64; ++sub) {
            if (wave == sub) {
                int nk = s_kept;
                if (alive)                                     // survivors added since phase A (lists are newest-first)
                    for (int k = head[lst]; k >= kept0 && alive; k = nxt[k])
                        if (iou_gt(kept[k], c, iouthr)) alive = false;
                // lanes of my list (ballot-built match mask), earlier and still alive
                unsigned long long same = ~0ull;
                if (by_class) {
#pragma unroll
                    for (int bit = 0; bit < 10; ++bit) {
                        const unsigned long long m = __ballot((lst >> bit) & 1);
                        same &= ((lst >> bit) & 1) ? m : ~m;
                    }
                }
                const unsigned long long amask = __ballot(alive);
                unsigned long long cand_sup = alive ? (same & amask & ((1ull << lane) - 1ull)) : 0ull;
                unsigned long long sup = 0;                    // earlier alive lanes of my class that overlap me
                const Cand* wb = wbox + sub * 64;
                while (cand_sup) {
                    const int i = __ffsll((long long)cand_sup) - 1;
                    cand_sup &= cand_sup - 1;
                    if (iou_gt(wb[i], c, iouthr)) sup |= 1ull << i;
                }
                // scalar scan in score order: lane j survives iff it is alive and none of its suppressors survived
                const unsigned int sup_lo = (unsigned int)sup, sup_hi = (unsigned int)(sup >> 32);
                unsigned long long keep = 0, todo = amask;
                int room = a.max_det - nk;
                while (todo != 0 && room > 0) {
                    const int j = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const unsigned long long sj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)sup_hi, j) << 32) |
                                                  (unsigned int)__builtin_amdgcn_readlane((int)sup_lo, j);
                    if ((sj & keep) == 0) { keep |= 1ull << j; --room; }
                }
                // survivors append themselves (parallel): kept list + class list
                if ((keep >> lane) & 1ull) {
                    const int slot = nk + __popcll(keep & ((1ull << lane) - 1ull));
                    kept[slot] = c;
                    nxt[slot] = (short)atomicExch(&head[lst], slot);
                }
                nk += __popcll(keep);
                if (lane == 0) s_kept = nk;
            }
            __syncthreads();
        }
        t_b += __builtin_readcyclecounter() - tb0;
    }
    if (b == 0 && tid == 0) {
        g_nms_dbg[0] = t_sorted - t0; g_nms_dbg[1] = t_a; g_nms_dbg[2] = t_b; g_nms_dbg[3] = __builtin_readcyclecounter() - t0;
        g_nms_dbg[4] = (unsigned long long)n; g_nms_dbg[5] = (unsigned long long)s_kept;
        g_nms_dbg[6] = (unsigned long long)s_wide; g_nms_dbg[7] = t_ld;
    }
    // ---- emit rows (x1,y1,x2,y2,conf,cls) of the survivors, un-offset boxes recomputed from the prediction (nms.py:21-28)
    const int nk = s_kept < a.max_det ? s_kept : a.max_det;
    for (int k = tid; k < nk; k += kSelT) {
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

__global__ __launch_bounds__(kSelT) void nms_select_kernel(const NmsArgs a) { nms_select_body(a, blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------------------
// ONE launch for the whole of non_max_suppression (maf_nms_ex flag MAF_NMS_SINGLE_LAUNCH: the latency path, small batches): every workgroup
// collects its share of the candidates; the workgroup that finishes an image LAST (a ticket per image, behind a device-scope fence: the other
// workgroups' keys and the count are visible to it) goes on and runs that image's sort + greedy selection (the one-workgroup form above,
// which takes any number of candidates).  No second launch, no host round trip, nobody spins: the other workgroups simply leave.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSelT) void nms_single_kernel(const NmsArgs a) {
    __shared__ int s_last;
    if (a.multi_label) nms_collect_multi_body(a); else nms_collect_best_body(a);
    __threadfence();                                           // release: this workgroup's keys before its ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&a.cnt[blockIdx.y * kCntStride + 2], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                           // acquire: everybody else's keys and the final count
    nms_select_body(a, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// Suppression-matrix path (images with <= kMaskN candidates — the normal case: ~2000 at conf 0.03).  The greedy scan of
// torchvision.ops.nms is sequential only in its *decisions*; all pair tests are independent.  So:
//   nms_sort_kernel   one workgroup per image: bitonic sort of the keys in LDS, candidates materialised in score order;
//   nms_mask_kernel   the whole chip: bit matrix M[i][j] = IoU(i, j) > thr for j > i, 64 x 64 blocks of the upper triangle;
//   nms_scan_kernel   one workgroup per image: walks the candidates 64 at a time — the 64 x 64 diagonal block decides the
//                     survivors of the block with a scalar bit loop (a survivor clears the bits it suppresses), then the rows
//                     of the survivors are OR-ed into the per-word "removed" state of all later blocks, lane = word.
// Every pair is tested with iou_gt on the class-offset boxes exactly as torchvision does, so no class reasoning is needed.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSortT = 1024;

// Per-class path.  Boxes of different classes are 4096 px apart (nms.py:94), so unless a box leaves its band only pairs of the SAME
// class can overlap: then the greedy scan splits into one short scan per class, all independent.  nms_sort_kernel decides per image:
// if the candidates' raw coordinates span less than 4095 px (no box can reach another class's band), no class holds more than
// kClsMax candidates and the key fields fit, it sorts by (class, score desc, box) and raises the image's flag; nms_cscan_kernel then
// runs the per-class scans on 16 wavefronts, sorts the survivors back into score order and emits the first max_det.  Every pair it
// does test goes through the same iou_gt on the same offset boxes, and every pair it skips has an empty intersection: the survivors
// are those of the full scan, bit for bit.  Images that do not qualify (agnostic, a dominant class, huge boxes) keep the matrix path.
constexpr int kClsMax = 256;          // longest per-class list the per-class path accepts
constexpr int kClsHist = 1024;        // classes it can count

__global__ __launch_bounds__(kSortT) void nms_sort_kernel(const NmsArgs a) {
    __shared__ unsigned long long lk[kMaskN];
    __shared__ int hist[kClsHist];
    __shared__ float red_lo[kSortT / 64], red_hi[kSortT / 64];
    __shared__ int s_bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    long long n64 = a.cnt[b * kCntStride];
    const long long cap = (long long)a.N * a.nc;
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    if (n == 0 || n > kMaskN) return;
    const unsigned long long* keys = a.keys + (size_t)b * a.capP;
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;
    int P = 64;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += kSortT) lk[i] = i < n ? keys[i] : ~0ull;
    bool cpath = a.cpath_ok != 0;
    if (cpath) {                                                   // uniform
        for (int i = tid; i < a.nc; i += kSortT) hist[i] = 0;
        if (tid == 0) s_bad = 0;
        __syncthreads();
        float lo = INFINITY, hi = -INFINITY; int bad = 0;
        for (int i = tid; i < n; i += kSortT) {
            const unsigned int flat = (unsigned int)(lk[i] & 0xffffffffu);
            const unsigned int box = flat / (unsigned int)a.nc;
            const int cls = (int)(flat - box * (unsigned int)a.nc);
            const float* r = pred + (size_t)box * no;
            const float cx = r[0], cy = r[1], w = r[2], h = r[3];
            const float x1 = cx - w / 2, y1 = cy - h / 2, x2 = cx + w / 2, y2 = cy + h / 2;
            const float mn = fminf(fminf(x1, y1), fminf(x2, y2)), mx = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
            if (!(mn > -INFINITY) || !(mx < INFINITY)) bad = 1;    // NaN / infinite coordinates: the literal all-pairs path
            lo = fminf(lo, mn); hi = fmaxf(hi, mx);
            if (atomicAdd(&hist[cls], 1) >= kClsMax) bad = 1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        if ((tid & 63) == 0) { red_lo[tid >> 6] = lo; red_hi[tid >> 6] = hi; }
        if (bad) s_bad = 1;
        __syncthreads();
        lo = red_lo[0]; hi = red_hi[0];
        for (int w = 1; w < kSortT / 64; ++w) { lo = fminf(lo, red_lo[w]); hi = fmaxf(hi, red_hi[w]); }
        cpath = !s_bad && (hi - lo) < kMaxWh - 1.0f;               // a pixel of slack for the rounding of the class offsets
        if (cpath)
            for (int i = tid; i < n; i += kSortT) {                // key: class | ~score | box  (same tie order as the flat index inside a class)
                const unsigned long long key = lk[i];
                const unsigned int flat = (unsigned int)(key & 0xffffffffu);
                const unsigned int box = flat / (unsigned int)a.nc;
                const unsigned long long cls = flat - box * (unsigned int)a.nc;
                lk[i] = (cls << (32 + a.bbits)) | ((key >> 32) << a.bbits) | box;
            }
    }
    if (tid == 0) a.cnt[b * kCntStride + 1] = cpath ? 1 : 0;
    __syncthreads();
    bitonic_sort(lk, P, tid, kSortT, true);
    Cand* out = a.cands + (size_t)b * kMaskN;
    for (int i = tid; i < n; i += kSortT) {
        const unsigned long long key = lk[i];
        unsigned int flat, sbits;
        if (cpath) {
            const unsigned int box = (unsigned int)(key & ((1ull << a.bbits) - 1ull));
            sbits = (unsigned int)(key >> a.bbits);
            flat = box * (unsigned int)a.nc + (unsigned int)(key >> (32 + a.bbits));
        } else {
            flat = (unsigned int)(key & 0xffffffffu);
            sbits = (unsigned int)(key >> 32);
        }
        const unsigned int box = flat / (unsigned int)a.nc;
        const int cls = (int)(flat - box * (unsigned int)a.nc);
        Cand c;
        make_cand(a, pred, no, box, cls, c);
        c.score = __uint_as_float(~sbits); c.flat = flat; c.pad = (unsigned int)cls;
        out[i] = c;
    }
}

__global__ __launch_bounds__(kSortT) void nms_cscan_kernel(const NmsArgs a) {
    __shared__ float bx1[kMaskN], by1[kMaskN], bx2[kMaskN], by2[kMaskN];     // the image's candidates, class-major (64 KiB)
    __shared__ unsigned long long skeys[kMaskN];                              // survivors as score-major keys (32 KiB)
    __shared__ unsigned char kept[kMaskN];
    __shared__ int seg[kClsHist + 1];
    __shared__ int s_ns;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!a.cnt[b * kCntStride + 1]) return;
    long long n64 = a.cnt[b * kCntStride];
    const long long cap = (long long)a.N * a.nc;
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    const Cand* cands = a.cands + (size_t)b * kMaskN;
    const IouThr iouthr = {a.iou, a.iou_m, a.iou_even, (float)a.iou_m};
    for (int i = tid; i < n; i += kSortT) {
        const Cand c = cands[i];
        bx1[i] = c.x1; by1[i] = c.y1; bx2[i] = c.x2; by2[i] = c.y2;
    }
    // seg[c] = first candidate of class >= c (lower bound on the class-major order)
    for (int c = tid; c <= a.nc; c += kSortT) {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)cands[mid].pad < c) lo = mid + 1; else hi = mid; }
        seg[c] = lo;
    }
    if (tid == 0) s_ns = 0;
    __syncthreads();
    auto box_at = [&](int i, Cand& c) {
        c.x1 = bx1[i]; c.y1 = by1[i]; c.x2 = bx2[i]; c.y2 = by2[i];
        c.area = (c.x2 - c.x1) * (c.y2 - c.y1);                                // = make_cand's area (same operands, same operation)
    };
    for (int cls = wave; cls < a.nc; cls += kSortT / 64) {
        const int s0 = seg[cls], e0 = seg[cls + 1];
        for (int base = s0; base < e0; base += 64) {
            const int idx = base + lane;
            const bool valid = idx < e0;
            Cand me; box_at(valid ? idx : s0, me);
            bool alive = valid;
            for (int k = s0; k < base; ++k) {                                  // survivors of the class's earlier blocks
                if (!kept[k]) continue;
                Cand kc; box_at(k, kc);
                if (alive && iou_gt(kc, me, iouthr)) alive = false;
            }
            unsigned long long sup_by = 0;                                     // earlier candidates of this block that overlap me
            const int bn = min(64, e0 - base);
            for (int i = 0; i < bn - 1; ++i) {
                Cand ic; box_at(base + i, ic);
                if (lane > i && valid && iou_gt(ic, me, iouthr)) sup_by |= 1ull << i;
            }
            unsigned long long todo = __ballot(alive), keep = 0;
            while (todo) {                                                     // a survivor removes the later candidates it overlaps
                const int i = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                keep |= 1ull << i;
                todo &= ~__ballot((sup_by >> i) & 1ull);
            }
            const bool k = (keep >> lane) & 1ull;
            if (valid) kept[idx] = k ? 1 : 0;
            if (k) {
                const Cand c = cands[idx];
                skeys[atomicAdd(&s_ns, 1)] = ((unsigned long long)(~__float_as_uint(c.score)) << 32) | c.flat;
            }
        }
    }
    __syncthreads();
    const int ns = s_ns;
    int P = 64;
    while (P < ns) P <<= 1;
    for (int i = ns + tid; i < P; i += kSortT) skeys[i] = ~0ull;
    __syncthreads();
    bitonic_sort(skeys, P, tid, kSortT, true);
    const int nk = ns < a.max_det ? ns : a.max_det;
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;
    float* orow = a.out_rows + (size_t)b * a.max_det * 6;
    long long* oidx = a.out_idx + (size_t)b * a.max_det;
    for (int k = tid; k < nk; k += kSortT) {
        const unsigned long long key = skeys[k];
        const unsigned int flat = (unsigned int)(key & 0xffffffffu);
        const unsigned int box = flat / (unsigned int)a.nc;
        const int cls = (int)(flat - box * (unsigned int)a.nc);
        const float* r = pred + (size_t)box * no;
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float* o = orow + (size_t)k * 6;
        o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2; o[4] = __uint_as_float(~(unsigned int)(key >> 32)); o[5] = (float)cls;
        oidx[k] = a.multi_label ? (long long)flat : (long long)box;
    }
    if (tid == 0) a.out_count[b] = nk;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsArgs a) {
    __shared__ Cand cb[64];
    const int b = blockIdx.y, lane = threadIdx.x;
    if (a.cnt[b * kCntStride + 1]) return;                     // the per-class path has this image
    long long n64 = a.cnt[b * kCntStride];
    const long long cap = (long long)a.N * a.nc;
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    if (n == 0 || n > kMaskN) return;
    const int nb = (n + 63) >> 6;
    const int npairs = nb * (nb + 1) / 2;
    const IouThr iouthr = {a.iou, a.iou_m, a.iou_even, (float)a.iou_m};
    const Cand* cands = a.cands + (size_t)b * kMaskN;
    unsigned long long* mask = a.mask + (size_t)b * kMaskN * kMaskW;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        // pair index -> (row block rb, column block cbk >= rb): rows of the upper triangle have nb, nb-1, ... entries
        int rb = 0, rem = pr;
        while (rem >= nb - rb) { rem -= nb - rb; ++rb; }
        const int cbk = rb + rem;
        const int i = rb * 64 + lane, j0 = cbk * 64;
        __syncthreads();
        if (j0 + lane < n) cb[lane] = cands[j0 + lane];
        __syncthreads();
        if (i < n) {
            const Cand me = cands[i];
            unsigned long long bits = 0;
            const int jn = min(64, n - j0);
            const int jlo = cbk == rb ? lane + 1 : 0;                // diagonal block: only later candidates
            for (int j = 0; j < jn; ++j) {                           // uniform trip count, no divergence in the common case
                const int r = iou_screen(me, cb[j], iouthr.m32);
                bool hit = r > 0;
                if (__any(r < 0)) {                                  // rare: some lane sits inside the screen's band
                    if (r < 0) hit = iou_gt(me, cb[j], iouthr);
                }
                bits |= (unsigned long long)(hit && j >= jlo) << j;
            }
            mask[(size_t)i * kMaskW + cbk] = bits;
        }
    }
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const NmsArgs a) {
    __shared__ unsigned long long rows[64][kMaskW + 1];        // the current block's matrix rows (+1: lanes walk a column conflict-free)
    __shared__ int kept_idx[kMaxDetCap];
    __shared__ int s_nk, s_done;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.cnt[b * kCntStride + 1]) return;                     // the per-class path has this image
    long long n64 = a.cnt[b * kCntStride];
    const long long cap = (long long)a.N * a.nc;
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    if (n == 0 || n > kMaskN) return;
    const int nb = (n + 63) >> 6;
    const unsigned long long* mask = a.mask + (size_t)b * kMaskN * kMaskW;
    if (tid == 0) { s_nk = 0; s_done = 0; }
    unsigned long long removed = 0;                            // wave 0: lane w = word w of the removed set
    // each thread stages 16 words of a block: row r = tid >> 2, words (tid & 3) * 16 .. +15 (only words >= blk are defined)
    unsigned long long pre0[16], pre1[16];                     // two blocks in flight: the matrix rows are read two blocks ahead
    auto load_blk = [&](int blk, unsigned long long (&pre)[16]) {
        const int r = tid >> 2, w0 = (tid & 3) * 16;
        const int i = blk * 64 + r;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int w = w0 + u;
            pre[u] = (i < n && w >= blk && w < nb) ? mask[(size_t)i * kMaskW + w] : 0ull;
        }
    };
    auto store_blk = [&](const unsigned long long (&pre)[16]) {
        const int r = tid >> 2, w0 = (tid & 3) * 16;
#pragma unroll
        for (int u = 0; u < 16; ++u) rows[r][w0 + u] = pre[u];
    };
    auto decide = [&](int blk) {                               // wave 0: survivors of block blk, then their rows into `removed`
        int nk = s_nk;
        const unsigned long long diag = rows[lane][blk];       // lane r: later candidates of this block that r suppresses
        const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
        const unsigned int rlo = (unsigned int)removed, rhi = (unsigned int)(removed >> 32);
        const unsigned long long rw = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)rhi, blk) << 32) |
                                      (unsigned int)__builtin_amdgcn_readlane((int)rlo, blk);
        const int nvalid = min(64, n - blk * 64);
        unsigned long long todo = ~rw & (nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull));
        unsigned long long keep = 0;
        int room = a.max_det - nk;
        while (todo != 0 && room > 0) {                        // scalar: a survivor clears the bits of the candidates it suppresses
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            keep |= 1ull << j;
            --room;
            const unsigned long long dj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane((int)dlo, j);
            todo &= ~dj;
        }
        if ((keep >> lane) & 1ull) kept_idx[nk + __popcll(keep & ((1ull << lane) - 1ull))] = blk * 64 + lane;
        nk += __popcll(keep);
        unsigned long long k2 = keep;                          // removed |= rows of the survivors (lane = word; words <= blk are dead)
        while (k2) {
            const int j = __ffsll((long long)k2) - 1;
            k2 &= k2 - 1;
            removed |= rows[j][lane];
        }
        if (lane == 0) { s_nk = nk; if (nk >= a.max_det) s_done = 1; }
    };
    load_blk(0, pre0);
    load_blk(1, pre1);
    for (int blk = 0; blk < nb; blk += 2) {
        __syncthreads();                                       // previous block's rows are no longer read
        store_blk(pre0);
        load_blk(blk + 2, pre0);
        __syncthreads();
        if (wave == 0) decide(blk);
        __syncthreads();
        if (s_done || blk + 1 >= nb) break;
        store_blk(pre1);
        load_blk(blk + 3, pre1);
        __syncthreads();
        if (wave == 0) decide(blk + 1);
        __syncthreads();
        if (s_done) break;
    }
    __syncthreads();
    // ---- emit rows (x1,y1,x2,y2,conf,cls) of the survivors, un-offset boxes recomputed from the prediction (nms.py:21-28)
    const int nk = s_nk < a.max_det ? s_nk : a.max_det;
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;
    const Cand* cands = a.cands + (size_t)b * kMaskN;
    float* orow = a.out_rows + (size_t)b * a.max_det * 6;
    long long* oidx = a.out_idx + (size_t)b * a.max_det;
    for (int k = tid; k < nk; k += 256) {
        const Cand& c = cands[kept_idx[k]];
        const unsigned int flat = c.flat;
        const unsigned int box = flat / (unsigned int)a.nc;
        const int cls = (int)(flat - box * (unsigned int)a.nc);
        const float* r = pred + (size_t)box * no;
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float* o = orow + (size_t)k * 6;
        o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2; o[4] = c.score; o[5] = (float)cls;
        oidx[k] = a.multi_label ? (long long)flat : (long long)box;
    }
    if (tid == 0) a.out_count[b] = nk;
}

// ---------------------------------------------------------------------------------------------------------------------
// Kept-list greedy scan of the all-pairs path (round 6; replaces nms_mask_kernel + nms_scan_kernel by default, which stay behind MAF_NMS_MATRIX for A/B):
// one workgroup of 16 waves per image walks the score-sorted candidates 64 at a time and tests a block only against the boxes KEPT so far (<= max_det) and
// against itself — n x kept / 2 + 64 n pairs (~0.3 M for 2 000 candidates, 300 kept) instead of the n^2 / 2 of the matrix (2.1 M), no n x n bit matrix in HBM,
// one launch less.  Per block: a wave tests every 16th kept box against the block's 64 candidates (lane = candidate; the kept box is a wave-uniform LDS
// broadcast, one ballot = its 64 verdicts) and every 16th row of the block's own 64 x 64 triangle the same way; wave 0 then runs the serial rule of the matrix
// scan (a survivor clears the later candidates it overlaps) and appends the survivors to the kept list.  Two blocks are in flight: while wave 0 decides block k
// the other 15 waves test block k + 1 against its triangle and the boxes kept before, and only the (<= 64) boxes block k adds are tested behind the barrier.
// Same predicate (iou_screen, iou_gt in the
// band), same (earlier, later) operand order, same greedy order: the survivors are those of the matrix path bit for bit (tests/test_gpu_model.py compares both).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kGreedyT = 1024;

__global__ __launch_bounds__(kGreedyT) void nms_greedy_kernel(const NmsArgs a) {
    __shared__ Cand kept[kMaxDetCap];                          // 32 KiB
    __shared__ Cand cb[2][64];                                 // the block being decided and the one being tested behind it
    __shared__ unsigned long long diag[2][64];                 // diag[.][i] bit j (> i): candidate i of the block overlaps candidate j
    __shared__ unsigned int dead_lo[2], dead_hi[2];            // candidates of the block a kept box overlaps
    __shared__ int s_nk;
    constexpr int NWV = kGreedyT / 64;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.cnt[b * kCntStride + 1]) return;                     // the per-class path has this image
    long long n64 = a.cnt[b * kCntStride];
    const long long cap = (long long)a.N * a.nc;
    if (n64 > cap) n64 = cap;
    const int n = (int)n64;
    if (n == 0 || n > kMaskN) return;                          // (no candidates: nms_select_kernel has written the zero count; more than kMaskN: it runs the image)
    const int nb = (n + 63) >> 6;
    const IouThr iouthr = {a.iou, a.iou_m, a.iou_even, (float)a.iou_m};
    const Cand* cands = a.cands + (size_t)b * kMaskN;
    const Cand zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
    // one (earlier box, this lane's candidate) verdict per lane, boxes as scalars in registers (a Cand handed on by reference went through scratch: 48 B of
    // private memory per lane and a global-memory round trip per test); the exact form only when some lane of the wave sits inside the screen's band
    // (wave-uniform branch)
    struct Box { float x1, y1, x2, y2, area; };
    auto ld = [&](const Cand* p) -> Box {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
        return Box{v[0], v[1], v[2], v[3], p->area};
    };
    // One (earlier box, this lane's candidate) verdict per lane.  The single-precision screen of iou_gt decides all but the pairs inside a band of 3e-7 x union around the
    // threshold; the loops below run the screen alone (two per-lane flags OR-ed over the boxes: hit, undecided) and only if some lane of the wave was left undecided by ANY box
    // — practically never — does the wave walk its boxes again with the exact form for those lanes.  (A pair the screen decided gets the same verdict from iou_gt, which
    // starts with the same screen: re-testing every box for the open lanes is consistent.)
    // v_min_f32 / v_max_f32 as they are: fminf / fmaxf put a canonicalising `v_max_f32 x, x, x` in front of every loaded operand (8 of the 43 vector instructions of two
    // tests).  Same result for every input but a SIGNALLING NaN coordinate (quiet NaNs — what arithmetic produces — give the other operand either way, and leave the pair
    // undecided: the exact form, which keeps fminf / fmaxf, then decides it).
    auto vmin = [](float x, float y) -> float { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto vmax = [](float x, float y) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto screen = [&](const Box k, const Box c, bool& hit, bool& open) {      // per-lane verdicts, OR-ed into the caller's (lane masks in scalar registers: one ballot behind the loop)
        const float w = vmax(0.f, vmin(k.x2, c.x2) - vmax(k.x1, c.x1)), h = vmax(0.f, vmin(k.y2, c.y2) - vmax(k.y1, c.y1));
        const float inter = w * h, uni = k.area + c.area - inter;
        const float d = __builtin_fmaf(-iouthr.m32, uni, inter);                 // = iou_screen
        const bool decided = uni > 0.f && fabsf(d) > 3e-7f * uni;
        hit = hit || (decided && d > 0.f);
        open = open || !decided;
    };
    auto exact = [&](const Box k, const Box c, unsigned long long lanes) -> unsigned long long {
        bool hit = false;
        if ((lanes >> lane) & 1ull) {
            const Cand kc = {k.x1, k.y1, k.x2, k.y2, k.area, 0.f, 0u, 0u}, cc = {c.x1, c.y1, c.x2, c.y2, c.area, 0.f, 0u, 0u};
            hit = iou_gt(kc, cc, iouthr);
        }
        return __ballot(hit) & lanes;
    };
    // tests of block `blk` (in cb[buf]) that do not need the verdict on the block before it: its own triangle, and the kept boxes [k0, k1); waves [w0, NWV) share them
    auto test_block = [&](int blk, int buf, int k0, int k1, bool triangle, int w0) {
        const int nvalid = min(64, n - blk * 64);
        const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        const Box me = ld(&cb[buf][lane]);
        const int nw = NWV - w0, w = wave - w0;
        if (w < 0) return;
        if (triangle)
            for (int i = w; i < nvalid; i += nw) {
                const unsigned long long later = valid & (i == 63 ? 0ull : (~0ull << (i + 1)));      // row i: candidate i against the later ones
                const Box bi = ld(&cb[buf][i]);
                bool h1 = false, o1 = false;
                screen(bi, me, h1, o1);
                unsigned long long hits = __ballot(h1) & later;
                const unsigned long long open = __ballot(o1) & later;
                if (open) hits |= exact(bi, me, open);
                if (lane == 0) diag[buf][i] = hits;
            }
        bool dl = false, ol = false;
        int k = k0 + w;
        for (; k + nw < k1; k += 2 * nw) {                     // two kept boxes per step: both LDS reads in flight before the first is used
            const Box ka = ld(&kept[k]), kb = ld(&kept[k + nw]);
            screen(ka, me, dl, ol);
            screen(kb, me, dl, ol);
        }
        if (k < k1) screen(ld(&kept[k]), me, dl, ol);
        unsigned long long dead = __ballot(dl) & valid;        // wave-uniform
        const unsigned long long open_any = __ballot(ol) & valid & ~dead;      // an undecided lane some other box already suppresses needs no second look
        if (open_any)
            for (int kk = k0 + w; kk < k1; kk += nw) dead |= exact(ld(&kept[kk]), me, open_any);
        if (lane == 0 && dead) { atomicOr(&dead_lo[buf], (unsigned int)dead); atomicOr(&dead_hi[buf], (unsigned int)(dead >> 32)); }
    };
    if (tid == 0) { s_nk = 0; dead_lo[0] = dead_hi[0] = dead_lo[1] = dead_hi[1] = 0u; }
    if (tid < 64) cb[0][tid] = tid < n ? cands[tid] : zero;
    else if (tid < 128) cb[1][tid - 64] = tid < n ? cands[tid] : zero;
    Cand nxt = zero;                                           // lanes 0..63 of wave 0: the block two ahead travels while the current ones are tested
    if (tid < 64 && 128 + tid < n) nxt = cands[128 + tid];
    __syncthreads();
    test_block(0, 0, 0, 0, true, 0);
    __syncthreads();
    int buf = 0;
#ifdef MAF_NMS_PROF
    unsigned long long t_dec = 0, t_testA = 0, t_barA = 0, t_B = 0, t_barB = 0, t_all0 = __builtin_readcyclecounter(), nblk = 0;
#endif
    for (int blk = 0; blk < nb; ++blk, buf ^= 1) {
        // here: cb[buf] = block blk with its triangle and its verdicts against EVERY kept box so far complete; cb[buf ^ 1] = block blk + 1, untested
        const int nk = s_nk;
        if (nk >= a.max_det) break;                            // uniform
        // ---- stage A: wave 0 decides block blk (the serial rule of the matrix scan); the other waves test block blk + 1 against its own triangle and the boxes kept BEFORE
#ifdef MAF_NMS_PROF
        const unsigned long long tA0 = __builtin_readcyclecounter();
        ++nblk;
#endif
        if (wave == 0) {
            const int nvalid = min(64, n - blk * 64);
            const Cand me = cb[buf][lane];
            const unsigned long long dg = lane < nvalid ? diag[buf][lane] : 0ull;
            const unsigned int dlo = (unsigned int)dg, dhi = (unsigned int)(dg >> 32);
            const unsigned long long dd = ((unsigned long long)dead_hi[buf] << 32) | dead_lo[buf];
            unsigned long long todo = ~dd & (nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull));
            unsigned long long keep = 0;
            int room = a.max_det - nk;
            while (todo != 0 && room > 0) {                    // one step per SURVIVOR: it clears the later candidates it overlaps
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                keep |= 1ull << j;
                --room;
                const unsigned long long dj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                                              (unsigned int)__builtin_amdgcn_readlane((int)dlo, j);
                todo &= ~dj;
            }
            if ((keep >> lane) & 1ull) kept[nk + __popcll(keep & ((1ull << lane) - 1ull))] = me;
            if (lane == 0) { s_nk = nk + __popcll(keep); dead_lo[buf] = 0u; dead_hi[buf] = 0u; }
        } else if (blk + 1 < nb) {
            test_block(blk + 1, buf ^ 1, 0, nk, true, 1);
        }
#ifdef MAF_NMS_PROF
        const unsigned long long tA1 = __builtin_readcyclecounter();
        if (wave == 0) t_dec += tA1 - tA0; else t_testA += tA1 - tA0;
#endif
        __syncthreads();
#ifdef MAF_NMS_PROF
        const unsigned long long tA2 = __builtin_readcyclecounter();
        t_barA += tA2 - tA1;
#endif
        if (blk + 1 >= nb) break;
        // ---- stage B: everybody tests block blk + 1 against the boxes block blk has just added; block blk + 2 takes the free buffer
        const int nk2 = s_nk;
        if (nk2 >= a.max_det) break;
        test_block(blk + 1, buf ^ 1, nk, nk2, false, 0);
        if (tid < 64) {
            cb[buf][tid] = nxt;
            const int i3 = (blk + 3) * 64 + tid;
            nxt = i3 < n ? cands[i3] : zero;
        }
#ifdef MAF_NMS_PROF
        const unsigned long long tB1 = __builtin_readcyclecounter();
        t_B += tB1 - tA2;
#endif
        __syncthreads();
#ifdef MAF_NMS_PROF
        t_barB += __builtin_readcyclecounter() - tB1;
#endif
    }
    __syncthreads();
#ifdef MAF_NMS_PROF
    if (b == 0 && lane == 0 && (wave == 0 || wave == 1)) {      // image 0: wave 0 = the decider, wave 1 = a tester
        const int o = wave * 4;
        g_nms_dbg[o + 0] = wave == 0 ? t_dec : t_testA; g_nms_dbg[o + 1] = t_barA; g_nms_dbg[o + 2] = t_B; g_nms_dbg[o + 3] = wave == 0 ? (__builtin_readcyclecounter() - t_all0) : ((nblk << 32) | (unsigned)s_nk);
        if (wave == 1) g_nms_dbg[o + 2] = t_barB;
    }
#endif
    // ---- emit rows (x1,y1,x2,y2,conf,cls) of the survivors, un-offset boxes recomputed from the prediction (nms.py:21-28)
    const int nk = s_nk < a.max_det ? s_nk : a.max_det;
    const int no = 5 + a.nc;
    const float* pred = a.pred + (size_t)b * a.N * no;
    float* orow = a.out_rows + (size_t)b * a.max_det * 6;
    long long* oidx = a.out_idx + (size_t)b * a.max_det;
    for (int k = tid; k < nk; k += kGreedyT) {
        const Cand& c = kept[k];
        const unsigned int flat = c.flat;
        const unsigned int box = flat / (unsigned int)a.nc;
        const int cls = (int)(flat - box * (unsigned int)a.nc);
        const float* r = pred + (size_t)box * no;
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float* o = orow + (size_t)k * 6;
        o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2; o[4] = c.score; o[5] = (float)cls;
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

extern "C" int maf_nms_debug(uint64_t* host8) {
    return maf_check_hip(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_nms_dbg), sizeof(unsigned long long) * 8), "hipMemcpyFromSymbol");
}

extern "C" int64_t maf_nms_workspace_bytes(int32_t B, int32_t N, int32_t nc) {
    if (B <= 0 || N <= 0 || nc <= 0) return 0;
    const long long capP = pow2ceil((long long)N * nc);
    return 256 + (long long)B * kCntStride * 4 + (long long)B * capP * 8 + (long long)B * kMaskN * 32 + (long long)B * kMaskN * kMaskW * 8;
}

extern "C" int maf_nms(const float* pred, int32_t B, int32_t N, int32_t nc, double conf_thres, double iou_thres,
                       const int32_t* classes, int32_t n_classes, int32_t agnostic, int32_t multi_label,
                       int32_t max_det, void* workspace, int64_t workspace_bytes,
                       float* out_rows, int64_t* out_idx, int32_t* out_count, maf_stream_t stream) {
    return maf_nms_ex(pred, B, N, nc, conf_thres, iou_thres, classes, n_classes, agnostic, multi_label, max_det, workspace, workspace_bytes,
                      out_rows, out_idx, out_count, 0, stream);
}

extern "C" int maf_nms_ex(const float* pred, int32_t B, int32_t N, int32_t nc, double conf_thres, double iou_thres,
                          const int32_t* classes, int32_t n_classes, int32_t agnostic, int32_t multi_label,
                          int32_t max_det, void* workspace, int64_t workspace_bytes,
                          float* out_rows, int64_t* out_idx, int32_t* out_count, int32_t flags, maf_stream_t stream) {
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
    {   // smallest fp32 strictly above the threshold, its predecessor, their midpoint.  The threshold is the double the caller passed
        // (torchvision's CPU kernel: `ovr > iou_threshold` with a double threshold) or, with MAF_NMS_FLOAT_THRESHOLD, that double rounded
        // to fp32 first (torchvision's CUDA kernel takes `float iou_threshold`): the two differ only for a pair whose fp32 IoU equals
        // fl32(thr) exactly AND fl32(thr) > thr (0.6, 0.7 ...: suppressed by the CPU rule, kept by the CUDA rule).
        float f = (float)iou_thres;
        const float tf = ((double)f > iou_thres && !(flags & MAF_NMS_FLOAT_THRESHOLD)) ? f : nextafterf(f, INFINITY);
        const float pf = nextafterf(tf, -INFINITY);
        a.iou_m = ((double)tf + (double)pf) * 0.5;
        uint32_t bits; memcpy(&bits, &tf, 4);
        a.iou_even = (bits & 1u) == 0;
    }
    a.classes = n_classes > 0 ? classes : nullptr; a.n_classes = n_classes;
    a.agnostic = agnostic; a.multi_label = (multi_label && nc > 1) ? 1 : 0;    // nms.py:57
    a.max_det = max_det;
    {   // per-class path: class and box index must fit beside the 32 score bits of the sort key
        int cb = 0, bb = 0;
        while ((1ll << cb) < nc) ++cb;
        while ((1ll << bb) < N) ++bb;
        a.bbits = bb;
        a.cpath_ok = (!agnostic && nc >= 8 && nc <= kClsHist && cb + bb <= 32) ? 1 : 0;
    }
    char* ws = static_cast<char*>(workspace);
    a.cnt = reinterpret_cast<int*>(ws);
    a.keys = reinterpret_cast<unsigned long long*>(ws + 256 + (long long)B * kCntStride * 4);
    a.capP = pow2ceil((long long)N * nc);
    a.cands = reinterpret_cast<Cand*>(ws + 256 + (long long)B * kCntStride * 4 + (long long)B * a.capP * 8);
    a.mask = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.cands) + (long long)B * kMaskN * 32);
    a.out_rows = out_rows; a.out_idx = reinterpret_cast<long long*>(out_idx); a.out_count = out_count;
    int rc = 0;
    const bool precollected = (flags & MAF_NMS_PRECOLLECTED) != 0;      // the forward pass (maf_engine_run_filtered) has filled counters and keys
    if (precollected) {
        MAF_REQUIRE(a.multi_label && !a.classes && conf_thres < 1.0 && !(flags & MAF_NMS_SINGLE_LAUNCH),
                    "nms: MAF_NMS_PRECOLLECTED needs multi_label (nc > 1), no class filter, conf_thres < 1, the multi-launch form");
        // (the other words of an image's counter line — per-class flag, ticket — still start from zero: the filter reset the whole line)
    } else {
        rc = maf_check_hip(hipMemsetAsync(a.cnt, 0, (size_t)B * kCntStride * 4, s), "nms memset");
        if (rc) return rc;
    }
    if (flags & MAF_NMS_SINGLE_LAUNCH) {                         // one launch: collect, then the last workgroup of every image selects (any candidate count)
        a.mask = nullptr;                                        // (nms_select_body: no suppression-matrix hand-over)
        int bx;
        if (a.multi_label) { const long long el = (long long)N * nc; bx = (int)((el + kChunk - 1) / kChunk < 256 ? (el + kChunk - 1) / kChunk : 256); }
        else bx = (N + 15) / 16 < 1024 ? (N + 15) / 16 : 1024;
        hipLaunchKernelGGL(nms_single_kernel, dim3(bx, B), dim3(kSelT), 0, s, a);
        return maf_check_hip(hipGetLastError(), "nms_single launch");
    }
    if (precollected) {
    } else if (a.multi_label) {
        const long long el = (long long)N * nc;
        const int bx = (int)((el + kChunk - 1) / kChunk < 256 ? (el + kChunk - 1) / kChunk : 256);
        hipLaunchKernelGGL(nms_collect_multi_kernel, dim3(bx, B), dim3(256), 0, s, a);
    } else {
        const int bx = (N + 15) / 16 < 1024 ? (N + 15) / 16 : 1024;      // 16 boxes per 256-thread block per iteration
        hipLaunchKernelGGL(nms_collect_best_kernel, dim3(bx, B), dim3(256), 0, s, a);
    }
    rc = maf_check_hip(hipGetLastError(), "nms_collect launch");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_select_kernel, dim3(B), dim3(kSelT), 0, s, a);        // images with > kMaskN candidates (others return at once)
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(kSortT), 0, s, a);
    if (a.cpath_ok) hipLaunchKernelGGL(nms_cscan_kernel, dim3(B), dim3(kSortT), 0, s, a);
    if (flags & MAF_NMS_MATRIX) {                                // A/B: the n x n suppression matrix (every pair tested by the whole chip) + the serial scan over its rows
        hipLaunchKernelGGL(nms_mask_kernel, dim3(kMaskWgs, B), dim3(64), 0, s, a);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(kGreedyT), 0, s, a);
    }
    return maf_check_hip(hipGetLastError(), "nms_select launch");
}
