// Launch-overlap probe: does a kernel launched WITHOUT the AQL barrier bit (hipExtLaunchKernel flag hipExtAnyOrderLaunch) start while its
// predecessors in the same stream are still running?  The engine uses that to run independent ops of the graph side by side inside ONE
// stream (no events, no extra streams); this probe measures it on the device at hand (maf_probe_anyorder, tools/anyorder_probe.py).
#include <hip/hip_ext.h>
#include "maf_common.h"

namespace {
__global__ void spin_kernel(long long cycles, int* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}
}  // namespace

// n launches of a `blocks`-workgroup kernel that spins `cycles` shader clocks each; flags = 0 (in-order) or hipExtAnyOrderLaunch for all but
// the first.  Returns the elapsed milliseconds between events around the n launches.
extern "C" int maf_probe_anyorder(void* stream, int n, int blocks, long long cycles, int flags, float* ms) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t a, b;
    int rc = maf_check_hip(hipEventCreate(&a), "hipEventCreate");
    if (!rc) rc = maf_check_hip(hipEventCreate(&b), "hipEventCreate");
    if (rc) return rc;
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, 1000ll, (int*)nullptr);      // warm-up
    rc = maf_check_hip(hipEventRecord(a, s), "hipEventRecord");
    for (int i = 0; i < n && !rc; ++i) {
        hipExtLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, i == 0 ? 0u : (unsigned)flags, cycles, (int*)nullptr);
        rc = maf_check_hip(hipGetLastError(), "spin launch");
    }
    if (!rc) rc = maf_check_hip(hipEventRecord(b, s), "hipEventRecord");
    if (!rc) rc = maf_check_hip(hipEventSynchronize(b), "hipEventSynchronize");
    if (!rc) rc = maf_check_hip(hipEventElapsedTime(ms, a, b), "hipEventElapsedTime");
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return rc;
}

// Vector-pipe throughput probe: cycles per wave64 instruction of v_fma_f32 / v_exp_f32 / v_rcp_f32 (8 independent chains per lane, one wave
// per SIMD x `waves` per SIMD).  kind: 0 fma, 1 exp, 2 rcp, 3 the SiLU sequence used in csrc/bottleneck.hip (exp2, add, rcp, mul).
namespace {
template <int KIND>
__global__ __launch_bounds__(256) void valu_probe_kernel(float* out, int iters, long long* cycles) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.001f * (float)(threadIdx.x + j);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
            else if (KIND == 1) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.25f;        // (+1 full-rate op to keep the value bounded)
            else if (KIND == 2) v[j] = __builtin_amdgcn_rcpf(v[j]) + 0.5f;          // (+1 full-rate op)
            else v[j] = v[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v[j])) + 0.7f;
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
}  // namespace

// -> shader clocks for `iters` iterations of 8 instructions (KIND 0) / 8 x (instruction + 1 full-rate op) (1, 2) / 8 SiLUs (3), measured on
// workgroup 0 with `wgs_per_cu` workgroups of 256 threads on every CU (1 -> one wave per SIMD, 2 -> two ...)
extern "C" int maf_probe_valu(void* stream, int kind, int iters, int wgs_per_cu, long long* host_cycles) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* out = nullptr; long long* cyc = nullptr;
    const int blocks = 256 * wgs_per_cu;
    int rc = maf_check_hip(hipMalloc(&out, (size_t)blocks * 256 * 4), "hipMalloc");
    if (!rc) rc = maf_check_hip(hipMalloc(&cyc, 8), "hipMalloc");
    if (rc) return rc;
    for (int rep = 0; rep < 2; ++rep) {
        switch (kind) {
            case 0: hipLaunchKernelGGL(valu_probe_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 1: hipLaunchKernelGGL(valu_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 2: hipLaunchKernelGGL(valu_probe_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            default: hipLaunchKernelGGL(valu_probe_kernel<3>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
        }
    }
    rc = maf_check_hip(hipStreamSynchronize(s), "sync");
    if (!rc) rc = maf_check_hip(hipMemcpy(host_cycles, cyc, 8, hipMemcpyDeviceToHost), "memcpy");
    (void)hipFree(out); (void)hipFree(cyc);
    return rc;
}

// ds_read_b64_tr_b16 semantics probe (tools/tr_probe.py): LDS half i holds the value i; lane l reads at byte address addr[l];
// out[l][0..3] = the four 16-bit values the lane receives.
namespace {
__global__ __launch_bounds__(64) void tr_probe_kernel(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    const unsigned a = (unsigned)(size_t)(&img[0]) + (unsigned)addr[threadIdx.x];
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v[0] & 0xffff);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v[0] >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v[1] & 0xffff);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v[1] >> 16);
}
}  // namespace

extern "C" int maf_probe_tr(void* stream, const int* addr_dev, unsigned short* out_dev) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), addr_dev, out_dev);
    return maf_check_hip(hipGetLastError(), "tr probe launch");
}
