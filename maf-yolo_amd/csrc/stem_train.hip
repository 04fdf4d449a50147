// Training-form stem: the two convolutions of backbone.0's RepVGGBlock over the IMAGE — 3x3 stride 2 pad 1 (rbr_dense) and 1x1 stride 2 (rbr_1x1),
// yolov6/layers/common.py:199-203, 219-224 (Cin = 3: MAF-YOLO-n.yaml:5) — in ONE launch, without bias or activation (their BatchNorms and the ReLU of the sum follow as
// csrc/bn_sum.hip's passes).  Training step: yolov6/core/engine.py:141-167, under autocast (fp16 operands, fp32 accumulation, fp16 results).
//
// As two launches of the generic kernels (the implicit GEMM with K = 9 taps x 8 padded channels = 72 and the 1x1 kernel reading pixel (2y, 2x)) the image — 32 x 640 x 640,
// zero-padded to 8 channels, 210 MB — was read twice and the matrix cores saw a K of 72 / 8: 168 + 104 us per step of MAF-YOLO-n at batch 32 (1.25 / 2.0 TB/s).  K = 27 is too thin for
// the matrix cores and the layer is bound by its bytes (210 MB in, 2 x 157 MB out), so — as csrc/stem.hip does for inference — a VALU direct conv: one thread owns 4 horizontally
// adjacent output pixels, reads the 3 x 9 input pixels under them as 16-byte NHWC8 loads (three useful channels each; the zero padding rides along) and keeps them as fp32
// pairs; the weights sit in LDS as [27 + 3][Cout] fp32, read as broadcast float4s shared by the 4 pixels, and are ROUNDED TO fp16 first (what autocast hands the reference's conv);
// the 1x1 branch is the centre pixel of the same registers against its own 3 x Cout weights.  Both results leave through an LDS staging area as coalesced 16-byte stores over the
// workgroup's contiguous NHWC span, one tensor after the other.  The parameters are read as they are ([Cout][3][3][3], [Cout][3][1][1] fp32): no packing step.
#include "maf_common.h"

namespace {

struct StemTrainArgs {
    const half_t* img;      // [B][Hin][Win][8] fp16, channels 3..7 zero
    const float* w3;        // [Cout][3][3][3]
    const float* w1;        // [Cout][3]
    half_t* z3; half_t* z1; // [B][H][W][Cout] dense
    int B, H, W, Hin, Win, Cout, img_stride;
};

template <int DUMMY>
__global__ __launch_bounds__(256) void stem_train_kernel(const StemTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [27][Cout] w3 (row = (c*3 + ky)*3 + kx), [3][Cout] w1, then the staging area
    const int nW = 27 * a.Cout, nW1 = 3 * a.Cout;
    for (int i = threadIdx.x; i < nW; i += blockDim.x) { const int k = i / a.Cout, co = i - k * a.Cout; smem[i] = (float)(half_t)a.w3[co * 27 + k]; }
    for (int i = threadIdx.x; i < nW1; i += blockDim.x) { const int c = i / a.Cout, co = i - c * a.Cout; smem[nW + i] = (float)(half_t)a.w1[co * 3 + c]; }
    __syncthreads();
    const int XQ = a.W >> 2;                                      // quads per output row (W % 4 == 0)
    const int total = a.B * a.H * XQ;
    const int tq = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = tq < total ? tq : total - 1;
    const int cpt = (4 * a.Cout * 2) >> 4;                        // 16-byte chunks per thread and tensor
    char* stage = reinterpret_cast<char*>(smem + nW + nW1) + (size_t)threadIdx.x * (cpt + 1) * 16;
    const int xq = t % XQ, t2 = t / XQ, y = t2 % a.H, b = t2 / a.H;
    const int x0 = xq * 4;
    const half_t* img = a.img + (size_t)b * a.Hin * a.Win * a.img_stride;
    f32x2_t in[9][5];                                             // [c*3 + ky][columns 2x0-1 .. 2x0+8 in register pairs]
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * y - 1 + ky;
        const bool rok = (unsigned)iy < (unsigned)a.Hin;
        const half_t* row = img + ((size_t)(rok ? iy : 0) * a.Win + 2 * x0) * a.img_stride;
#pragma unroll
        for (int i = 0; i < 9; ++i) {                             // column 2x0 - 1 + i
            const bool ok = rok && (i > 0 || x0 > 0);
            const half8_t h = *reinterpret_cast<const half8_t*>(row + (ok ? i - 1 : 0) * a.img_stride);
#pragma unroll
            for (int c = 0; c < 3; ++c) in[c * 3 + ky][i >> 1][i & 1] = ok ? (float)h[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) in[c * 3 + ky][4][1] = 0.f;
    }
    const float* w1s = smem + nW;
    // ---- two passes over the output channels: the 3x3 branch, then the 1x1 branch; each staged and stored on its own
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
#pragma unroll 1
        for (int c0 = 0; c0 < a.Cout; c0 += 8) {
            f32x2_t acc[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] = f32x2_t{0.f, 0.f};
            if (which == 0) {
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    __builtin_amdgcn_sched_barrier(0);            // one (channel, ky) row of weights in flight (see csrc/stem.hip)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int k = r * 3 + kx;
                        const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0]);
                        const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(&smem[k * a.Cout + c0 + 4]);
                        const f32x2_t wk[4] = {f32x2_t{w0[0], w0[1]}, f32x2_t{w0[2], w0[3]}, f32x2_t{w1[0], w1[1]}, f32x2_t{w1[2], w1[3]}};
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const f32x2_t vv = in[r][(2 * p + kx) >> 1];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (kx & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[p][j]) : "v"(vv), "v"(wk[j]));
                                else        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[p][j]) : "v"(vv), "v"(wk[j]));
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {                     // pixel (2y, 2(x0 + p)) = row ky 1, column index 2p + 1: the high register of pair p
                    const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(&w1s[c * a.Cout + c0]);
                    const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(&w1s[c * a.Cout + c0 + 4]);
                    const f32x2_t wk[4] = {f32x2_t{w0[0], w0[1]}, f32x2_t{w0[2], w0[3]}, f32x2_t{w1[0], w1[1]}, f32x2_t{w1[2], w1[3]}};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f32x2_t vv = in[c * 3 + 1][p];
#pragma unroll
                        for (int j = 0; j < 4; ++j) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[p][j]) : "v"(vv), "v"(wk[j]));
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                half8_t v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (half_t)acc[p][j >> 1][j & 1];
                *reinterpret_cast<half8_t*>(stage + ((size_t)p * a.Cout + c0) * 2) = v;
            }
        }
        __syncthreads();
        // thread t of the grid owns pixels 4t .. 4t+3, so the workgroup's output is ONE contiguous span
        const int first = blockIdx.x * blockDim.x;
        const int nthr = total - first < (int)blockDim.x ? total - first : (int)blockDim.x;
        const char* sbase = reinterpret_cast<const char*>(smem + nW + nW1);
        char* gbase = reinterpret_cast<char*>((which == 0 ? a.z3 : a.z1) + (size_t)first * 4 * a.Cout);
        for (int q = threadIdx.x; q < nthr * cpt; q += blockDim.x) {
            const int thr = q / cpt, off = q - thr * cpt;
            *reinterpret_cast<f32x4_t*>(gbase + (size_t)q * 16) = *reinterpret_cast<const f32x4_t*>(sbase + ((size_t)thr * (cpt + 1) + off) * 16);
        }
        __syncthreads();
    }
}

}  // namespace

// z3 = conv3x3 s2 p1 (img, w3), z1 = conv1x1 s2 (img, w1): img [B][Hin][Win][8] fp16 NHWC (pixel stride img_stride >= 8 halfs, channels 3.. zero or ignored: only 0..2 are read),
// w3 [Cout][3][3][3] and w1 [Cout][3] fp32 (the parameters themselves), z3 / z1 dense [B][Hin/2][Win/2][Cout] fp16.  Cout a multiple of 8, Win a multiple of 8, even Hin.
extern "C" int maf_stem_train(const void* img, int32_t img_stride, int32_t B, int32_t Hin, int32_t Win, const float* w3, const float* w1, int32_t Cout,
                              void* z3, void* z1, maf_stream_t stream) {
    MAF_REQUIRE(img && w3 && w1 && z3 && z1 && B > 0, "stem_train: null pointer / empty batch");
    MAF_REQUIRE(img_stride >= 8 && img_stride % 8 == 0, "stem_train: the image is NHWC with 16-byte pixels (3 channels padded to 8)");
    MAF_REQUIRE(Hin > 0 && Hin % 2 == 0 && Win > 0 && Win % 8 == 0, "stem_train: even image height, width a multiple of 8");
    MAF_REQUIRE(Cout % 8 == 0 && Cout > 0 && Cout <= 96, "stem_train: Cout a multiple of 8, at most 96");
    StemTrainArgs a;
    a.img = static_cast<const half_t*>(img); a.w3 = w3; a.w1 = w1; a.z3 = static_cast<half_t*>(z3); a.z1 = static_cast<half_t*>(z1);
    a.B = B; a.Hin = Hin; a.Win = Win; a.H = Hin / 2; a.W = Win / 2; a.Cout = Cout; a.img_stride = img_stride;
    const int M = B * a.H * (a.W / 4);
    const int slot = 4 * Cout * 2 + 16;
    int threads = (44 * 1024 / slot) / 64 * 64;
    threads = threads > 256 ? 256 : threads < 64 ? 64 : threads;
    const size_t sh = (size_t)(30 * Cout) * sizeof(float) + (size_t)threads * slot;
    MAF_REQUIRE(sh <= 64 * 1024, "stem_train: Cout too large for the LDS staging area");
    hipLaunchKernelGGL((stem_train_kernel<0>), dim3(maf_cdiv(M, threads)), dim3(threads), sh, static_cast<hipStream_t>(stream), a);
    return maf_check_hip(hipGetLastError(), "stem_train launch");
}

namespace {

// [B][3][H][W] planar (fp32 or fp16) -> [B][H][W][8] fp16, channels 3..7 zero: one pass (torch's copy_ into the channel slice of the NHWC8 buffer ran as a strided fp32 copy,
// a cast, a fill and a strided 16-bit copy: 186 us per step of batch 32 at 640 x 640, in front of the first kernel of the forward)
template <typename TI>
__global__ __launch_bounds__(256) void image_nhwc8_kernel(const TI* __restrict__ x, half_t* __restrict__ out, long long plane, long long total4) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;           // 4 consecutive pixels of one image
    if (q >= total4) return;
    const long long per = plane >> 2, b = q / per, p = (q - b * per) << 2;
    const TI* src = x + b * 3 * plane + p;
    float v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if constexpr (sizeof(TI) == 4) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(src + c * plane);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = t[i];
        } else {
            typedef half_t h4 __attribute__((ext_vector_type(4)));
            const h4 t = *reinterpret_cast<const h4*>(src + c * plane);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = (float)t[i];
        }
    }
    half_t* dst = out + (b * plane + p) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        half8_t o = (half8_t)(half_t)0;
        o[0] = (half_t)v[0][i]; o[1] = (half_t)v[1][i]; o[2] = (half_t)v[2][i];
        *reinterpret_cast<half8_t*>(dst + i * 8) = o;
    }
}

}  // namespace

// The training step's input staging: a contiguous NCHW image batch [B][3][H][W] (dtype MAF_F32 or MAF_F16; H * W a multiple of 4) into the NHWC8 fp16 buffer the train-form
// stem reads (maf_stem_train, the stem's weight-gradient kernels): out [B][H][W][8], channels 3..7 zero.
extern "C" int maf_image_to_nhwc8(const void* x, int32_t dtype, int32_t B, int32_t H, int32_t W, void* out, maf_stream_t stream) {
    MAF_REQUIRE(x && out && B > 0 && H > 0 && W > 0, "image_to_nhwc8: bad arguments");
    MAF_REQUIRE(dtype == MAF_F32 || dtype == MAF_F16, "image_to_nhwc8: the image is fp32 or fp16");
    const long long plane = (long long)H * W;
    MAF_REQUIRE(plane % 4 == 0, "image_to_nhwc8: H * W must be a multiple of 4");
    const long long total4 = (long long)B * (plane >> 2);
    const dim3 g((unsigned)((total4 + 255) / 256)), b(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F32) hipLaunchKernelGGL((image_nhwc8_kernel<float>), g, b, 0, s, static_cast<const float*>(x), static_cast<half_t*>(out), plane, total4);
    else hipLaunchKernelGGL((image_nhwc8_kernel<half_t>), g, b, 0, s, static_cast<const half_t*>(x), static_cast<half_t*>(out), plane, total4);
    return maf_check_hip(hipGetLastError(), "image_to_nhwc8 launch");
}
