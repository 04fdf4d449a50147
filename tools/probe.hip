// Launch-overlap probe: does a kernel launched WITHOUT the AQL barrier bit (hipExtLaunchKernel flag hipExtAnyOrderLaunch) start while its
// predecessors in the same stream are still running?  The engine uses that to run independent ops of the graph side by side inside ONE
// stream (no events, no extra streams); this probe measures it on the device at hand (maf_probe_anyorder; result in DESIGN.md 5a: accepted and ignored on gfx950).
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
    const uint32_t sw = 0x2c002800u + (uint32_t)(iters & 1);                         // wave-uniform, not a literal: lives in a scalar register
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
            else if (KIND == 1) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.25f;        // (+1 full-rate op to keep the value bounded)
            else if (KIND == 2) v[j] = __builtin_amdgcn_rcpf(v[j]) + 0.5f;          // (+1 full-rate op)
            else if (KIND == 3) v[j] = v[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v[j])) + 0.7f;
            else if (KIND == 4) {                                                    // v_fma_mix_f32 (f16 sources, fp32 accumulate): the depth-wise kernels' multiply-add
                const uint32_t a = 0x38003c00u + (uint32_t)j, b = 0x3c003800u;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(v[j]) : "v"(a), "v"(b));
            } else if (KIND == 5) {                                                  // v_dot2_f32_f16: two multiply-adds into one fp32 accumulator
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                v[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, 0x38003c00u + (uint32_t)j), __builtin_bit_cast(h2, 0x2c002800u), v[j] * 0.0f + v[j], false);
            } else if (KIND == 6) {                                                  // v_pk_fma_f16
                uint32_t x = __builtin_bit_cast(uint32_t, v[j]);
                asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(0x38003800u), "v"(0x34003400u));
                v[j] = __builtin_bit_cast(float, x);
            } else if (KIND == 7) {                                                  // v_pk_fma_f32 on a register pair
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {v[j], v[(j + 1) & 7]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"((f2){0.999f, 0.999f}), "v"((f2){0.001f, 0.001f}));
                v[j] = x[0];
            } else if (KIND == 8) {                                                  // v_dot2c_f32_f16 (accumulates in place)
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[j]) : "v"(0x38003c00u + (uint32_t)j), "v"(0x2c002800u));
            } else if (KIND == 9) {                                                  // v_fma_mix_f32 with a SCALAR-register operand (csrc/dwconv_sw.hip)
                const uint32_t a = 0x38003c00u + (uint32_t)j;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(v[j]) : "v"(a), "s"(sw + j));
            } else if (KIND == 10) {                                                 // v_fma_f32 with a scalar-register operand
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "s"(__builtin_bit_cast(float, 0x3f7fbe77u + (sw & 1u))), "v"(0.001f));
            } else {                                                                 // KIND 11: v_dot2c_f32_f16 with a scalar-register operand
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[j]) : "s"(sw + j), "v"(0x38003c00u + (uint32_t)j));
            }
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
extern "C" int maf_probe_valu_ms(void* stream, int kind, int iters, int wgs_per_cu, long long* host_cycles, float* ms) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* out = nullptr; long long* cyc = nullptr;
    const int blocks = 256 * wgs_per_cu;
    int rc = maf_check_hip(hipMalloc(&out, (size_t)blocks * 256 * 4), "hipMalloc");
    if (!rc) rc = maf_check_hip(hipMalloc(&cyc, 8), "hipMalloc");
    if (rc) return rc;
    hipEvent_t e0, e1;
    rc = maf_check_hip(hipEventCreate(&e0), "hipEventCreate");
    if (!rc) rc = maf_check_hip(hipEventCreate(&e1), "hipEventCreate");
    if (rc) return rc;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) (void)hipEventRecord(e0, s);
        switch (kind) {
            case 0: hipLaunchKernelGGL(valu_probe_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 1: hipLaunchKernelGGL(valu_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 2: hipLaunchKernelGGL(valu_probe_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 3: hipLaunchKernelGGL(valu_probe_kernel<3>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 4: hipLaunchKernelGGL(valu_probe_kernel<4>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 5: hipLaunchKernelGGL(valu_probe_kernel<5>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 6: hipLaunchKernelGGL(valu_probe_kernel<6>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 7: hipLaunchKernelGGL(valu_probe_kernel<7>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 8: hipLaunchKernelGGL(valu_probe_kernel<8>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 9: hipLaunchKernelGGL(valu_probe_kernel<9>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            case 10: hipLaunchKernelGGL(valu_probe_kernel<10>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
            default: hipLaunchKernelGGL(valu_probe_kernel<11>, dim3(blocks), dim3(256), 0, s, out, iters, cyc); break;
        }
    }
    (void)hipEventRecord(e1, s);
    rc = maf_check_hip(hipStreamSynchronize(s), "sync");
    if (!rc && ms) rc = maf_check_hip(hipEventElapsedTime(ms, e0, e1), "hipEventElapsedTime");      // the second launch alone
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!rc) rc = maf_check_hip(hipMemcpy(host_cycles, cyc, 8, hipMemcpyDeviceToHost), "memcpy");
    (void)hipFree(out); (void)hipFree(cyc);
    return rc;
}

extern "C" int maf_probe_valu(void* stream, int kind, int iters, int wgs_per_cu, long long* host_cycles) {
    return maf_probe_valu_ms(stream, kind, iters, wgs_per_cu, host_cycles, nullptr);
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


// PMC calibration kernels (tools/pmc_calibrate.py): stream a buffer of known size with a known access shape so that rocprofv3's FETCH_SIZE /
// WRITE_SIZE can be compared with a byte count (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a 16-byte-per-lane streaming read on
// gfx950; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//   kind 0 / 1 / 2   copy with 16- / 8- / 4-byte lanes (reads = writes = bytes)
//   kind 3           16-byte lanes reading every OTHER 16-byte chunk (a stride-2 gather: reads bytes / 2, writes bytes / 2)
//   kind 4           16-byte reads, 2-byte writes (one half per lane, contiguous across lanes: reads bytes, writes bytes / 8)
//   kind 5           64-byte segments: 4 lanes x 16 bytes of every other 64 bytes (the half-line shape of an MFMA activation fragment)
namespace {
template <int KIND>
__global__ __launch_bounds__(256) void pmc_cal_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long bytes) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthr = (long long)gridDim.x * 256;
    if (KIND == 0) { for (long long i = tid; i < bytes / 16; i += nthr) reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[i]; }
    else if (KIND == 1) { for (long long i = tid; i < bytes / 8; i += nthr) reinterpret_cast<u32x2_t*>(dst)[i] = reinterpret_cast<const u32x2_t*>(src)[i]; }
    else if (KIND == 2) { for (long long i = tid; i < bytes / 4; i += nthr) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i]; }
    else if (KIND == 3) { for (long long i = tid; i < bytes / 32; i += nthr) reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[2 * i]; }
    else if (KIND == 4) { for (long long i = tid; i < bytes / 16; i += nthr) { const u32x4_t v = reinterpret_cast<const u32x4_t*>(src)[i]; reinterpret_cast<unsigned short*>(dst)[i] = (unsigned short)(v[0] ^ v[3]); } }
    else { for (long long i = tid; i < bytes / 32; i += nthr) { const long long seg = i >> 2, l = i & 3; reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[seg * 8 + l]; } }
}
}  // namespace

extern "C" int maf_probe_pmc_copy(void* stream, int kind, const void* src, void* dst, long long bytes) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned char* a = static_cast<const unsigned char*>(src);
    unsigned char* b = static_cast<unsigned char*>(dst);
    const dim3 g(256 * 8), t(256);
    switch (kind) {
        case 0: hipLaunchKernelGGL(pmc_cal_kernel<0>, g, t, 0, s, a, b, bytes); break;
        case 1: hipLaunchKernelGGL(pmc_cal_kernel<1>, g, t, 0, s, a, b, bytes); break;
        case 2: hipLaunchKernelGGL(pmc_cal_kernel<2>, g, t, 0, s, a, b, bytes); break;
        case 3: hipLaunchKernelGGL(pmc_cal_kernel<3>, g, t, 0, s, a, b, bytes); break;
        case 4: hipLaunchKernelGGL(pmc_cal_kernel<4>, g, t, 0, s, a, b, bytes); break;
        case 5: hipLaunchKernelGGL(pmc_cal_kernel<5>, g, t, 0, s, a, b, bytes); break;
        default: maf_set_error("pmc_copy: kind 0..5"); return MAF_E_ARG;
    }
    return maf_check_hip(hipGetLastError(), "pmc_cal launch");
}
