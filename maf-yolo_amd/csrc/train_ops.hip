// Training-side helpers (the backward of the 1x1 and depth-wise conv families).
//
// The train-form graph keeps conv and BatchNorm separate (yolov6/layers/common.py:46-47, 219-224, 3024-3031), so the
// convolutions run without a fused epilogue and their backward splits into
//   dgrad  = the SAME forward kernels on transformed weights: 1x1 -> conv_mfma with W^T, depth-wise -> dwconv_tile with the
//            spatially flipped kernel (both weight transforms are done on the device by the pack kernels below, one launch);
//   wgrad  = 1x1: a plain TN GEMM  dW = dY^T X  (left to hipBLASLt through torch.mm, as a plain library GEMM);
//            depth-wise: dw_wgrad_kernel below (per-channel k*k reductions over all pixels; MIOpen's weak spot).
#include "maf_common.h"

namespace {

// fp32 [Cout][Cin] (or its transpose) -> MFMA B-fragment order of conv_mfma (see pack.py / conv_mfma.inc.h):
//   packed[tile = nt*CT + ct][step][lane = g*16 + p][j] = W[chan = nt*16*CT + p*CT + ct][k = step*KS + g*CH + j]
template <typename T, int CH>
__global__ __launch_bounds__(256) void pack_w1x1_kernel(const float* __restrict__ w, int Cout, int Cin, int transpose, int CT, int steps,
                                                        T* __restrict__ out, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % CH);
    long long t = e / CH;
    const int lane = (int)(t % 64); t /= 64;
    const int step = (int)(t % steps);
    const int tile = (int)(t / steps);
    const int g = lane >> 4, p = lane & 15;
    const int nt = tile / CT, ct = tile - nt * CT;
    const int chan = nt * 16 * CT + p * CT + ct;
    const int k = step * (4 * CH) + g * CH + j;
    // logical matrix Wl[chan][k]: transpose == 0 -> W[chan][k] (rows = Cout); transpose == 1 -> W[k][chan] (dgrad: rows = Cin)
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    float v = 0.f;
    if (chan < rows && k < cols) v = transpose ? w[(size_t)k * Cin + chan] : w[(size_t)chan * Cin + k];
    out[e] = (T)v;
}

// fp32 [C][k*k] -> [k*k][C] in T, optionally spatially flipped (dgrad of a stride-1 "same" depth-wise conv)
template <typename T>
__global__ __launch_bounds__(256) void pack_dw_kernel(const float* __restrict__ w, int C, int kk, int flip, T* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * kk) return;
    const int c = e % C, tap = e / C;
    out[e] = (T)w[(size_t)c * kk + (flip ? kk - 1 - tap : tap)];
}

// Depth-wise weight gradient: dW[c][ky][kx] = sum_{b,y,x} dY[b,y,x,c] * X[b,y+ky-P,x+kx-P,c]   (zero padding).
// Workgroup = one TH x TW tile of one image x one block of CB channels: the X halo tile and the dY tile are staged in
// LDS once; each lane owns (16-byte channel group, tap) pairs, accumulates over the tile's pixels in fp32 and adds its
// partial sums to dW with fp32 atomics (dW is zeroed by the caller).
struct DwWgArgs {
    const void* x; const void* dy; float* dw;
    int B, H, W, C, x_stride, dy_stride, TH, TW, CB, tilesX, tilesY, nCB;
};

template <typename T, typename V, int N, int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwWgArgs a) {
    constexpr int P = K / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    V* xt = reinterpret_cast<V*>(smem_raw);
    int t = blockIdx.x;
    const int cb = t % a.nCB; t /= a.nCB;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * a.TH, x0 = tx * a.TW, c0 = cb * a.CB;
    const int CGB = min(a.CB, a.C - c0) / N;
    const int PS = a.CB / N + 1;                               // LDS pixel stride in vectors (+16 B pad)
    const int RH = a.TH + K - 1, RW = a.TW + K - 1;
    V* dt = xt + RH * RW * PS;                                 // [TH*TW][PS]
    const int tid = threadIdx.x;
    const T* xin = static_cast<const T*>(a.x) + c0;
    const T* dyin = static_cast<const T*>(a.dy) + c0;
    for (int idx = tid; idx < RH * RW * CGB; idx += 256) {
        const int cgi = idx % CGB, p = idx / CGB;
        const int rx = p % RW, ry = p / RW;
        const int iy = y0 - P + ry, ix = x0 - P + rx;
        V v = (V)(T)0;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = *reinterpret_cast<const V*>(xin + ((size_t)((size_t)b * a.H + iy) * a.W + ix) * a.x_stride + cgi * N);
        xt[p * PS + cgi] = v;
    }
    for (int idx = tid; idx < a.TH * a.TW * CGB; idx += 256) {
        const int cgi = idx % CGB, p = idx / CGB;
        const int rx = p % a.TW, ry = p / a.TW;
        const int oy = y0 + ry, ox = x0 + rx;
        V v = (V)(T)0;
        if (oy < a.H && ox < a.W)
            v = *reinterpret_cast<const V*>(dyin + ((size_t)((size_t)b * a.H + oy) * a.W + ox) * a.dy_stride + cgi * N);
        dt[p * PS + cgi] = v;
    }
    __syncthreads();
    for (int pr = tid; pr < CGB * K * K; pr += 256) {
        const int cgi = pr % CGB, tap = pr / CGB;
        const int ky = tap / K, kx = tap - ky * K;
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        for (int ry = 0; ry < a.TH; ++ry) {
            const V* xr = xt + ((ry + ky) * RW + kx) * PS + cgi;
            const V* dr = dt + (ry * a.TW) * PS + cgi;
            for (int rx = 0; rx < a.TW; ++rx) {
                const V xv = xr[rx * PS], dv = dr[rx * PS];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = __builtin_fmaf((float)xv[j], (float)dv[j], acc[j]);
            }
        }
        float* o = a.dw + (size_t)(c0 + cgi * N) * (K * K) + tap;
#pragma unroll
        for (int j = 0; j < N; ++j) atomicAdd(o + (size_t)j * (K * K), acc[j]);
    }
}

template <typename T, typename V, int N>
int launch_dw_wgrad(DwWgArgs& a, int k, hipStream_t s) {
    a.TH = min(8, a.H); a.TW = min(16, a.W);
    a.CB = min(8 * N, (a.C + N - 1) / N * N);
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    const size_t lds = ((size_t)(a.TH + k - 1) * (a.TW + k - 1) + (size_t)a.TH * a.TW) * (a.CB / N + 1) * 16;
    const dim3 g(a.B * a.tilesY * a.tilesX * a.nCB), b(256);
    switch (k) {
        case 3: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 3>), g, b, lds, s, a); break;
        case 5: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 5>), g, b, lds, s, a); break;
        case 7: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 7>), g, b, lds, s, a); break;
        case 9: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 9>), g, b, lds, s, a); break;
        default: maf_set_error("dw_wgrad: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
    return maf_check_hip(hipGetLastError(), "dw_wgrad launch");
}

}  // namespace

extern "C" int maf_pack_w1x1(const float* w, int32_t Cout, int32_t Cin, int32_t transpose, int32_t dtype, int32_t tile_c,
                             void* out, maf_stream_t stream) {
    MAF_REQUIRE(w && out && Cout > 0 && Cin > 0, "pack_w1x1: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "pack_w1x1: dtype must be f16/f32");
    MAF_REQUIRE(tile_c == 2 || tile_c == 4 || tile_c == 6 || tile_c == 8, "pack_w1x1: tile_c in {2,4,6,8}");
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int CH = dtype == MAF_F16 ? 8 : 4, KS = 4 * CH;
    const int steps = maf_cdiv(cols, KS), tiles = maf_cdiv(rows, 16 * tile_c) * tile_c;
    const long long total = (long long)tiles * steps * 64 * CH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((pack_w1x1_kernel<half_t, 8>), g, b, 0, s, w, Cout, Cin, transpose, tile_c, steps, static_cast<half_t*>(out), total);
    else hipLaunchKernelGGL((pack_w1x1_kernel<float, 4>), g, b, 0, s, w, Cout, Cin, transpose, tile_c, steps, static_cast<float*>(out), total);
    return maf_check_hip(hipGetLastError(), "pack_w1x1 launch");
}

extern "C" int64_t maf_pack_w1x1_bytes(int32_t Cout, int32_t Cin, int32_t transpose, int32_t dtype, int32_t tile_c) {
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int CH = dtype == MAF_F16 ? 8 : 4, KS = 4 * CH;
    return (int64_t)maf_cdiv(rows, 16 * tile_c) * tile_c * maf_cdiv(cols, KS) * 64 * 16;
}

extern "C" int maf_pack_dw(const float* w, int32_t C, int32_t k, int32_t flip, int32_t dtype, void* out, maf_stream_t stream) {
    MAF_REQUIRE(w && out && C > 0 && k > 0, "pack_dw: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "pack_dw: dtype must be f16/f32");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g(maf_cdiv(C * k * k, 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((pack_dw_kernel<half_t>), g, b, 0, s, w, C, k * k, flip, static_cast<half_t*>(out));
    else hipLaunchKernelGGL((pack_dw_kernel<float>), g, b, 0, s, w, C, k * k, flip, static_cast<float*>(out));
    return maf_check_hip(hipGetLastError(), "pack_dw launch");
}

extern "C" int maf_dw_wgrad(const void* x, int32_t x_stride, const void* dy, int32_t dy_stride, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t k, int32_t dtype, float* dw, maf_stream_t stream) {
    MAF_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && C > 0, "dw_wgrad: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "dw_wgrad: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && x_stride % N == 0 && dy_stride % N == 0, "dw_wgrad: C and strides must be multiples of the 16-byte channel group");
    DwWgArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.B = B; a.H = H; a.W = W; a.C = C; a.x_stride = x_stride; a.dy_stride = dy_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) return launch_dw_wgrad<half_t, half8_t, 8>(a, k, s);
    return launch_dw_wgrad<float, f32x4_t, 4>(a, k, s);
}
