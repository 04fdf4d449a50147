// Shared device/host helpers for libmafyolo_hip (gfx950 only: wave64, MFMA 16x16x32 f16 / 16x16x4 f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mafyolo_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

void maf_set_error(const std::string& msg);
int maf_check_hip(hipError_t e, const char* what);

#define MAF_REQUIRE(cond, msg)                          \
    do {                                                \
        if (!(cond)) {                                  \
            maf_set_error(std::string(msg));            \
            return MAF_E_ARG;                           \
        }                                               \
    } while (0)

// ---- per-kind launchers (each in its own .hip file) ----
int maf_launch_conv_mfma(const maf_op_t* op, hipStream_t s);   // CONV1X1, CONV3X3S2
int maf_launch_stem(const maf_op_t* op, hipStream_t s);
int maf_launch_dwconv(const maf_op_t* op, hipStream_t s);
int maf_launch_sppf_pool(const maf_op_t* op, hipStream_t s);
int maf_launch_decode(const maf_op_t* op, hipStream_t s);
int maf_launch_bottleneck(const maf_op_t* op, hipStream_t s);
int maf_launch_conv1dw(const maf_op_t* op, hipStream_t s);
int maf_launch_head_tail(const maf_op_t* op, hipStream_t s);
int maf_launch_stem2(const maf_op_t* op, hipStream_t s);
int maf_launch_conv3s2_lds(const maf_op_t* op, hipStream_t s);
int maf_launch_conv3s2_wreg(const maf_op_t* op, hipStream_t s);
// depth-wise weight gradient on the matrix cores (dw_wgrad_mfma.hip); MAF_E_UNSUPPORTED = shape not covered, nothing launched
int maf_dw_wgrad_mfma(const void* x, int x_stride, const void* dy, int dy_stride, int B, int H, int W, int C, int k, float* dw, int replicas, hipStream_t s);

// ---- device helpers ----
template <int ACT>
__device__ __forceinline__ float maf_act(float x) {
    if (ACT == MAF_ACT_RELU) return x > 0.f ? x : 0.f;
    if (ACT == MAF_ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));      // v_exp + v_rcp (1 ulp), no IEEE divide
    if (ACT == MAF_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return x;
}
__device__ __forceinline__ float maf_act_rt(float x, int act) {
    switch (act) {
        case MAF_ACT_RELU: return x > 0.f ? x : 0.f;
        case MAF_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
        case MAF_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-x));
        default: return x;
    }
}

static inline int maf_cdiv(int a, int b) { return (a + b - 1) / b; }
