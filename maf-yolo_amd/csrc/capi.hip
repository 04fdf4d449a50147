// C-ABI glue: error reporting, op dispatch, the plan executor (engine) and HIP-event timers.
#include <vector>
#include <string>
#include "maf_common.h"

static thread_local std::string g_err;

void maf_set_error(const std::string& msg) { g_err = msg; }

int maf_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return MAF_E_HIP;
}

extern "C" const char* maf_last_error(void) { return g_err.c_str(); }
// hipMemsetAsync(p, 0, bytes) on `stream`: accumulation buffers of kernels that run on a stream the framework's allocator / fill kernels are
// not on (the weight-gradient side stream of the training layers)
extern "C" int maf_zero(void* p, int64_t bytes, maf_stream_t stream) {
    MAF_REQUIRE(p && bytes >= 0, "maf_zero: bad arguments");
    return maf_check_hip(hipMemsetAsync(p, 0, (size_t)bytes, static_cast<hipStream_t>(stream)), "maf_zero");
}

extern "C" int maf_version(void) { return 200; }
extern "C" int maf_op_size(void) { return (int)sizeof(maf_op_t); }

extern "C" int maf_op_launch(const maf_op_t* op, maf_stream_t stream) {
    if (!op) { maf_set_error("maf_op_launch: null op"); return MAF_E_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (op->kind) {
        case MAF_OP_STEM: return maf_launch_stem(op, s);
        case MAF_OP_CONV1X1:
        case MAF_OP_CONV3X3S2:
        case MAF_OP_CONV3X3S2_DGRAD: return maf_launch_conv_mfma(op, s);
        case MAF_OP_DWCONV: return maf_launch_dwconv(op, s);
        case MAF_OP_SPPF_POOL: return maf_launch_sppf_pool(op, s);
        case MAF_OP_DECODE: return maf_launch_decode(op, s);
        case MAF_OP_BOTTLENECK: return maf_launch_bottleneck(op, s);
        case MAF_OP_CONV1DW: return maf_launch_conv1dw(op, s);
        case MAF_OP_HEADTAIL: return maf_launch_head_tail(op, s);
        case MAF_OP_STEM2: return maf_launch_stem2(op, s);
        default: maf_set_error("maf_op_launch: unknown op kind"); return MAF_E_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------
// Engine: the flattened Model.forward (yolov6/models/yolo.py:186-201) as a static launch list over
// a caller-owned activation arena.  No allocation, no host sync; optional hipGraph replay.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxLanes = 8;

struct maf_engine {
    std::vector<maf_op_t> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const void* g_image = nullptr;
    void* g_pred = nullptr;
    // multi-lane execution: side streams (lane 1..), one event per op that is read from another lane, fork/join events
    int n_lanes = 1;
    hipStream_t side[kMaxLanes] = {};
    std::vector<hipEvent_t> done;          // done[i] valid iff signal[i]
    std::vector<char> signal;
    hipEvent_t fork = nullptr, join[kMaxLanes] = {};
    bool lanes_ready = false;
};

static int engine_prepare_lanes(maf_engine* e) {
    if (e->lanes_ready) return 0;
    const size_t n = e->ops.size();
    e->signal.assign(n, 0);
    e->done.assign(n, nullptr);
    e->n_lanes = 1;
    for (size_t i = 0; i < n; ++i) {
        const maf_op_t& o = e->ops[i];
        if (o.lane < 0 || o.lane >= kMaxLanes) { maf_set_error("engine: op lane out of range (0..7)"); return MAF_E_ARG; }
        if (o.n_wait < 0 || o.n_wait > 8) { maf_set_error("engine: n_wait out of range (0..8)"); return MAF_E_ARG; }
        if (o.lane + 1 > e->n_lanes) e->n_lanes = o.lane + 1;
        for (int k = 0; k < o.n_wait; ++k) {
            if (o.wait[k] < 0 || (size_t)o.wait[k] >= i) { maf_set_error("engine: an op may only wait for earlier ops"); return MAF_E_ARG; }
            e->signal[o.wait[k]] = 1;
        }
    }
    int rc = 0;
    for (int l = 1; l < e->n_lanes && !rc; ++l) {
        rc = maf_check_hip(hipStreamCreateWithFlags(&e->side[l], hipStreamNonBlocking), "hipStreamCreate");
        if (!rc) rc = maf_check_hip(hipEventCreateWithFlags(&e->join[l], hipEventDisableTiming), "hipEventCreate");
    }
    if (!rc && e->n_lanes > 1) rc = maf_check_hip(hipEventCreateWithFlags(&e->fork, hipEventDisableTiming), "hipEventCreate");
    for (size_t i = 0; i < n && !rc; ++i)
        if (e->signal[i]) rc = maf_check_hip(hipEventCreateWithFlags(&e->done[i], hipEventDisableTiming), "hipEventCreate");
    if (!rc) e->lanes_ready = true;
    return rc;
}

extern "C" int maf_engine_create(const maf_op_t* ops, int32_t n_ops, maf_engine_t** out) {
    if (!ops || n_ops <= 0 || !out) { maf_set_error("maf_engine_create: bad arguments"); return MAF_E_ARG; }
    maf_engine* e = new maf_engine();
    e->ops.assign(ops, ops + n_ops);
    *out = e;
    return 0;
}

extern "C" int maf_engine_num_ops(const maf_engine_t* e) { return e ? (int)e->ops.size() : 0; }

// where maf_nms_ex keeps its candidate lists inside a workspace (csrc/nms.hip: counters | keys | ...)
static long long nms_cap_pow2(long long v) { long long p = 1; while (p < v) p <<= 1; return p; }

static int engine_launch_all(maf_engine* e, const void* image, void* pred, hipStream_t s, void* cand_ws = nullptr, float cand_conf = 0.f) {
    int rc = engine_prepare_lanes(e);
    if (rc) return rc;
    const bool multi = e->n_lanes > 1;
    int* cand_cnt = nullptr; unsigned long long* cand_keys = nullptr; long long cand_cap = 0;
    if (cand_ws) {                                           // candidate filter in the head tails: every level must end in one
        int B = 0, A = 0, nc = 0, tails = 0;
        for (const maf_op_t& o : e->ops) {
            if (o.kind == MAF_OP_DECODE) { maf_set_error("maf_engine_run_filtered: a level of this plan is decoded by MAF_OP_DECODE (no fused tail)"); return MAF_E_UNSUPPORTED; }
            if (o.kind == MAF_OP_HEADTAIL) { B = o.B; A = o.Win; nc = o.nc; ++tails; }
        }
        if (!tails) { maf_set_error("maf_engine_run_filtered: the plan has no MAF_OP_HEADTAIL"); return MAF_E_UNSUPPORTED; }
        cand_cnt = static_cast<int*>(cand_ws);
        cand_keys = reinterpret_cast<unsigned long long*>(static_cast<char*>(cand_ws) + 256 + (long long)B * MAF_NMS_CNT_STRIDE * 4);
        cand_cap = nms_cap_pow2((long long)A * nc);          // the key capacity maf_nms_ex derives from (N, nc) for the same workspace
        // the counter reset is queued BEFORE the fork event: a head tail on a side lane is then ordered behind it by the fork itself, not
        // only through its data dependencies on lane-0 ops
        rc = maf_check_hip(hipMemsetAsync(cand_cnt, 0, (size_t)B * MAF_NMS_CNT_STRIDE * 4, s), "candidate counter reset");
        if (rc) return rc;
    }
    if (multi) {                                             // fork: the side lanes start after everything already queued on s
        rc = maf_check_hip(hipEventRecord(e->fork, s), "hipEventRecord(fork)");
        for (int l = 1; l < e->n_lanes && !rc; ++l) rc = maf_check_hip(hipStreamWaitEvent(e->side[l], e->fork, 0), "hipStreamWaitEvent(fork)");
        if (rc) return rc;
    }
    for (size_t i = 0; i < e->ops.size(); ++i) {
        maf_op_t op = e->ops[i];
        if ((op.kind == MAF_OP_STEM || op.kind == MAF_OP_STEM2) && image) op.src[0].ptr = image;
        if ((op.kind == MAF_OP_DECODE || op.kind == MAF_OP_HEADTAIL) && pred) op.out = pred;
        if (op.kind == MAF_OP_HEADTAIL && cand_ws) { op.aux[1] = cand_cnt; op.aux[2] = cand_keys; op.lvl_h[0] = (int32_t)cand_cap; op.lvl_stride[1] = cand_conf; }
        hipStream_t st = op.lane == 0 ? s : e->side[op.lane];
        for (int k = 0; k < op.n_wait && !rc; ++k) rc = maf_check_hip(hipStreamWaitEvent(st, e->done[op.wait[k]], 0), "hipStreamWaitEvent");
        if (!rc) rc = maf_op_launch(&op, st);
        if (!rc && e->signal[i]) rc = maf_check_hip(hipEventRecord(e->done[i], st), "hipEventRecord");
        if (rc) {
            g_err = "op " + std::to_string(i) + ": " + g_err;
            return rc;
        }
    }
    if (multi) {                                             // join: whatever the caller queues next on s sees the whole forward
        for (int l = 1; l < e->n_lanes && !rc; ++l) {
            rc = maf_check_hip(hipEventRecord(e->join[l], e->side[l]), "hipEventRecord(join)");
            if (!rc) rc = maf_check_hip(hipStreamWaitEvent(s, e->join[l], 0), "hipStreamWaitEvent(join)");
        }
    }
    return rc;
}

extern "C" int maf_engine_run(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream) {
    if (!e) { maf_set_error("maf_engine_run: null engine"); return MAF_E_ARG; }
    return engine_launch_all(e, image, pred, static_cast<hipStream_t>(stream));
}

extern "C" int maf_engine_run_filtered(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream, void* nms_workspace, double conf_thres) {
    if (!e || !nms_workspace || !pred) { maf_set_error("maf_engine_run_filtered: null argument"); return MAF_E_ARG; }
    if (!(conf_thres >= 0.0 && conf_thres < 1.0)) { maf_set_error("maf_engine_run_filtered: conf_thres must be in [0, 1)"); return MAF_E_ARG; }
    return engine_launch_all(e, image, pred, static_cast<hipStream_t>(stream), nms_workspace, (float)conf_thres);
}

extern "C" int maf_engine_run_graph(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream) {
    if (!e) { maf_set_error("maf_engine_run_graph: null engine"); return MAF_E_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (e->exec && (e->g_image != image || e->g_pred != pred)) {
        (void)hipGraphExecDestroy(e->exec); (void)hipGraphDestroy(e->graph);
        e->exec = nullptr; e->graph = nullptr;
    }
    if (!e->exec) {
        int rc = maf_check_hip(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
        if (rc) return rc;
        rc = engine_launch_all(e, image, pred, s);
        hipError_t ce = hipStreamEndCapture(s, &e->graph);
        if (rc) return rc;
        rc = maf_check_hip(ce, "hipStreamEndCapture");
        if (rc) return rc;
        rc = maf_check_hip(hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        if (rc) return rc;
        e->g_image = image; e->g_pred = pred;
    }
    return maf_check_hip(hipGraphLaunch(e->exec, s), "hipGraphLaunch");
}

extern "C" int maf_engine_run_timed(maf_engine_t* e, const void* image, void* pred, maf_stream_t stream, float* ms_per_op) {
    if (!e || !ms_per_op) { maf_set_error("maf_engine_run_timed: bad arguments"); return MAF_E_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = e->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    int rc = 0;
    for (size_t i = 0; i <= n && !rc; ++i) rc = maf_check_hip(hipEventCreate(&ev[i]), "hipEventCreate");
    if (!rc) rc = maf_check_hip(hipEventRecord(ev[0], s), "hipEventRecord");
    for (size_t i = 0; i < n && !rc; ++i) {
        maf_op_t op = e->ops[i];
        if ((op.kind == MAF_OP_STEM || op.kind == MAF_OP_STEM2) && image) op.src[0].ptr = image;
        if ((op.kind == MAF_OP_DECODE || op.kind == MAF_OP_HEADTAIL) && pred) op.out = pred;
        rc = maf_op_launch(&op, s);
        if (!rc) rc = maf_check_hip(hipEventRecord(ev[i + 1], s), "hipEventRecord");
    }
    if (!rc) rc = maf_check_hip(hipEventSynchronize(ev[n]), "hipEventSynchronize");
    for (size_t i = 0; i < n && !rc; ++i) rc = maf_check_hip(hipEventElapsedTime(&ms_per_op[i], ev[i], ev[i + 1]), "hipEventElapsedTime");
    for (size_t i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

extern "C" void maf_engine_destroy(maf_engine_t* e) {
    if (!e) return;
    if (e->exec) (void)hipGraphExecDestroy(e->exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    for (int l = 1; l < kMaxLanes; ++l) {
        if (e->join[l]) (void)hipEventDestroy(e->join[l]);
        if (e->side[l]) (void)hipStreamDestroy(e->side[l]);
    }
    if (e->fork) (void)hipEventDestroy(e->fork);
    for (hipEvent_t ev : e->done) if (ev) (void)hipEventDestroy(ev);
    delete e;
}

// ---------------------------------------------------------------------------------------------
// HIP-event timer on an explicit stream (bench.py measures the stream the kernels run on)
// ---------------------------------------------------------------------------------------------
struct maf_timer { hipEvent_t a, b; };

extern "C" int maf_timer_create(void** t) {
    if (!t) return MAF_E_ARG;
    maf_timer* m = new maf_timer();
    int rc = maf_check_hip(hipEventCreate(&m->a), "hipEventCreate");
    if (!rc) rc = maf_check_hip(hipEventCreate(&m->b), "hipEventCreate");
    if (rc) { delete m; return rc; }
    *t = m;
    return 0;
}
extern "C" int maf_timer_start(void* t, maf_stream_t s) { return maf_check_hip(hipEventRecord(static_cast<maf_timer*>(t)->a, static_cast<hipStream_t>(s)), "hipEventRecord"); }
extern "C" int maf_timer_stop(void* t, maf_stream_t s) { return maf_check_hip(hipEventRecord(static_cast<maf_timer*>(t)->b, static_cast<hipStream_t>(s)), "hipEventRecord"); }
extern "C" int maf_timer_elapsed_ms(void* t, float* ms) {
    maf_timer* m = static_cast<maf_timer*>(t);
    int rc = maf_check_hip(hipEventSynchronize(m->b), "hipEventSynchronize");
    if (rc) return rc;
    return maf_check_hip(hipEventElapsedTime(ms, m->a, m->b), "hipEventElapsedTime");
}
extern "C" void maf_timer_destroy(void* t) {
    maf_timer* m = static_cast<maf_timer*>(t);
    if (!m) return;
    (void)hipEventDestroy(m->a); (void)hipEventDestroy(m->b);
    delete m;
}

// A stream whose kernels may only run on the compute units of a bit mask (hipExtStreamCreateWithCUMask; bit i = CU i in the driver's numbering, which
// deals consecutive bits round-robin to the 8 XCDs): the NMS of batch i on a slice of the chip while the forward of batch i + 1 has the rest.
extern "C" int maf_stream_create_masked(const uint32_t* mask_words, int32_t n_words, maf_stream_t* out) {
    if (!mask_words || n_words <= 0 || !out) { maf_set_error("stream_create_masked: bad arguments"); return MAF_E_ARG; }
    hipStream_t s = nullptr;
    if (int rc = maf_check_hip(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words), "hipExtStreamCreateWithCUMask")) return rc;
    *out = s;
    return 0;
}
extern "C" int maf_stream_destroy(maf_stream_t s) { return maf_check_hip(hipStreamDestroy(static_cast<hipStream_t>(s)), "hipStreamDestroy"); }
