// Step tape (include/mafyolo_hip.h, "Step tape"): replays a recorded list of C-ABI calls of one training step with one call from the host.
// The reference issues a train step op by op from Python (yolov6/core/engine.py:141-167: model forward, loss, scaler.scale(loss).backward()); the
// train-form graph here is ~900 short kernels, and issuing them one by one from autograd Functions costs the host as long as the step takes
// on the device.  A record is (entry point, stream selector, argument words); the entry points are the ordinary exported functions of this
// library, called through their own prototypes — the argument conversion below is derived by the compiler from the declarations in the
// header, so a changed signature cannot go unnoticed.  Host code only: no kernel lives here.
#include <cstring>
#include <type_traits>
#include <utility>
#include "maf_common.h"

namespace {

template <class T>
inline T arg_from_word(uint64_t v) {
    if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(v);
    else if constexpr (std::is_same_v<T, float>) { const uint32_t b = (uint32_t)v; float f; memcpy(&f, &b, 4); return f; }
    else if constexpr (std::is_same_v<T, double>) { double d; memcpy(&d, &v, 8); return d; }
    else return static_cast<T>(v);
}

template <class... A, size_t... I>
inline int call_words(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) { return fn(arg_from_word<A>(a[I])...); }

template <class... A>
inline int call_fn(int (*fn)(A...), const uint64_t* a) {
    static_assert(sizeof...(A) <= MAF_TAPE_MAX_ARGS, "entry point has more arguments than a tape record holds");
    return call_words(fn, a, std::index_sequence_for<A...>{});
}

template <class... A>
constexpr int nargs_of(int (*)(A...)) { return (int)sizeof...(A); }

struct Entry {
    const char* name;
    int (*call)(const uint64_t*);
    int nargs;
};

#define MAF_TAPE_ENTRY(f) { #f, +[](const uint64_t* a) -> int { return call_fn(&f, a); }, nargs_of(&f) }

// every entry takes its maf_stream_t LAST (the executor overwrites that word with the stream the record selects)
const Entry kEntries[] = {
    MAF_TAPE_ENTRY(maf_op_launch),
    MAF_TAPE_ENTRY(maf_pack_batch),
    MAF_TAPE_ENTRY(maf_pack_w1x1),
    MAF_TAPE_ENTRY(maf_pack_dw),
    MAF_TAPE_ENTRY(maf_bn_forward_ex),
    MAF_TAPE_ENTRY(maf_bn_backward_acc),
    MAF_TAPE_ENTRY(maf_bn_stats),
    MAF_TAPE_ENTRY(maf_bn_sum_forward),
    MAF_TAPE_ENTRY(maf_bn_sum_forward_stats),
    MAF_TAPE_ENTRY(maf_bn_sum_backward),
    MAF_TAPE_ENTRY(maf_dw_branches),
    MAF_TAPE_ENTRY(maf_dw_branches_stats),
    MAF_TAPE_ENTRY(maf_dw_wgrad),
    MAF_TAPE_ENTRY(maf_dw_wgrad31),
    MAF_TAPE_ENTRY(maf_stem_train),
    MAF_TAPE_ENTRY(maf_conv_wgrad),
    MAF_TAPE_ENTRY(maf_grad_fold),
    MAF_TAPE_ENTRY(maf_zero),
    MAF_TAPE_ENTRY(maf_colsum),
    MAF_TAPE_ENTRY(maf_add_sub2),
    MAF_TAPE_ENTRY(maf_nhwc_sum),
    MAF_TAPE_ENTRY(maf_maxpool_forward),
    MAF_TAPE_ENTRY(maf_maxpool_backward),
    MAF_TAPE_ENTRY(maf_upsample2x_forward),
    MAF_TAPE_ENTRY(maf_upsample2x_backward),
    MAF_TAPE_ENTRY(maf_detect_join),
    MAF_TAPE_ENTRY(maf_detect_join_backward),
};
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));

}  // namespace

extern "C" int32_t maf_tape_fn_id(const char* name) {
    if (!name) return -1000;
    if (!strcmp(name, "maf_stream_fork")) return MAF_TAPE_FORK;
    for (int i = 0; i < kNumEntries; ++i)
        if (!strcmp(name, kEntries[i].name)) return i;
    return -1000;
}

extern "C" int32_t maf_tape_fn_nargs(int32_t fn) {
    if (fn == MAF_TAPE_FORK) return 2;
    return fn >= 0 && fn < kNumEntries ? kEntries[fn].nargs : -1;
}

extern "C" int32_t maf_tape_rec_size(void) { return (int32_t)sizeof(maf_tape_rec_t); }

// records [first, last) in order; streams[0] = the main stream, [1] = the weight-gradient stream, [2..] = lanes of independent branches; the first failing
// record's index goes to *failed_at (maf_last_error holds its message)
extern "C" int maf_tape_run(const maf_tape_rec_t* recs, int32_t first, int32_t last, const maf_stream_t* streams, int32_t n_streams, int32_t* failed_at) {
    MAF_REQUIRE(recs && first >= 0 && last >= first && streams && n_streams >= 1, "tape_run: bad arguments");
    uint64_t a[MAF_TAPE_MAX_ARGS];
    for (int32_t i = first; i < last; ++i) {
        const maf_tape_rec_t& r = recs[i];
        int rc;
        if (r.fn == MAF_TAPE_FORK) {                        // a[0] -> a[1]: stream a[1] waits for what stream a[0] holds
            if (r.a[0] >= (uint64_t)n_streams || r.a[1] >= (uint64_t)n_streams) { maf_set_error("tape_run: stream index out of range in a fork record"); rc = MAF_E_ARG; }
            else rc = maf_stream_fork(streams[r.a[0]], streams[r.a[1]]);
        } else if (r.fn >= 0 && r.fn < kNumEntries && r.stream >= 0 && r.stream < n_streams) {
            const Entry& e = kEntries[r.fn];
            memcpy(a, r.a, sizeof(uint64_t) * e.nargs);
            a[e.nargs - 1] = reinterpret_cast<uint64_t>(streams[r.stream]);
            rc = e.call(a);
        } else {
            maf_set_error("tape_run: unknown entry point or stream in a record");
            rc = MAF_E_ARG;
        }
        if (rc) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return 0;
}

extern "C" int maf_tape_toggle(const maf_tape_toggle_t* t, int32_t n) {
    MAF_REQUIRE((t || n == 0) && n >= 0, "tape_toggle: bad arguments");
    for (int32_t i = 0; i < n; ++i) {
        if (t[i].width == 8) *static_cast<uint64_t*>(t[i].addr) ^= t[i].mask;
        else if (t[i].width == 4) *static_cast<uint32_t*>(t[i].addr) ^= (uint32_t)t[i].mask;
        else { maf_set_error("tape_toggle: width must be 4 or 8"); return MAF_E_ARG; }
    }
    return 0;
}
