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
