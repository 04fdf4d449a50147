// Training-side helpers (the backward of the 1x1 and depth-wise conv families).
//
// The train-form graph keeps conv and BatchNorm separate (yolov6/layers/common.py:46-47, 219-224, 3024-3031), so the
// convolutions run without a fused epilogue and their backward splits into
//   dgrad  = the SAME forward kernels on transformed weights: 1x1 -> conv_mfma with W^T, depth-wise -> dwconv_tile with the
//            spatially flipped kernel (both weight transforms are done on the device by the pack kernels below, one launch);
//   wgrad  = 1x1: a plain TN GEMM  dW = dY^T X  (left to hipBLASLt through torch.mm, as a plain library GEMM);
//            depth-wise: dw_wgrad_kernel below (per-channel k*k reductions over all pixels; MIOpen's weak spot).
#include <cstdio>
#include <cstdlib>
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

// All weight transforms of one training step in ONE launch (maf_pack_batch): the step needs every dense weight in fragment order twice
// (forward, and transposed for the data gradient) and every depth-wise kernel twice (as is, and flipped) — ~250 launches of a few
// microseconds each when issued per layer.  The descriptors live in device memory (the parameter addresses of a model do not change
// from step to step); block b finds its descriptor by binary search over the descriptors' first-block numbers.
__global__ __launch_bounds__(256) void pack_batch_kernel(const maf_pack_desc_t* __restrict__ descs, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                                 // last descriptor whose block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const maf_pack_desc_t d = descs[lo];
    const float* __restrict__ w = static_cast<const float*>(d.src);
    const long long e0 = ((long long)(blockIdx.x - d.block0) * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long e = e0 + q;
        if (e >= d.total) return;
        float v = 0.f;
        if (d.kind == 0) {                                            // dense: see pack_w1x1_kernel; K runs tap-major, every tap padded to Kp
            const int CH = d.dtype == MAF_F16 ? 8 : 4;
            const int j = (int)(e % CH);
            long long t = e / CH;
            const int lane = (int)(t % 64); t /= 64;
            const int step = (int)(t % d.steps);
            const int tile = (int)(t / d.steps);
            const int g = lane >> 4, p = lane & 15;
            const int nt = tile / d.CT, ct = tile - nt * d.CT;
            const int chan = nt * 16 * d.CT + p * d.CT + ct;
            const int kcol = step * (4 * CH) + g * CH + j;
            const int tap = kcol / d.Kp, k = kcol - tap * d.Kp;
            const int rows = d.transpose ? d.Cin : d.Cout, cols = d.transpose ? d.Cout : d.Cin;
            if (chan < rows && k < cols && tap < d.taps)
                v = d.transpose ? w[((size_t)k * d.Cin + chan) * d.taps + tap] : w[((size_t)chan * d.Cin + k) * d.taps + tap];
        } else if (d.kind == 2) {                                     // fp32 vector, zero-padded to `total` (a prediction conv's bias on the conv's channel tile)
            v = e < d.Cout ? w[e] : 0.f;
        } else {                                                      // depth-wise: [C][kk] -> [kk][C], optionally flipped
            const int c = (int)(e % d.Cout), tap = (int)(e / d.Cout);
            v = w[(size_t)c * d.taps + (d.flip ? d.taps - 1 - tap : tap)];
        }
        if (d.dtype == MAF_F16) static_cast<half_t*>(d.dst)[e] = (half_t)v;
        else static_cast<float*>(d.dst)[e] = v;
    }
}

// ModelEMA.update (yolov6/utils/ema.py:25-37) for every floating-point state_dict entry in ONE launch: e = e * d + (1 - d) * m, the two products
// and the sum rounded separately (the reference's `v *= d; v += (1 - d) * msd[k]` — no fused multiply-add), fp32.
__global__ __launch_bounds__(256) void ema_update_kernel(const maf_ema_desc_t* __restrict__ descs, int n, float d, float omd) {
#pragma clang fp contract(off)                                        // (HIP's __fmul_rn / __fadd_rn are inline operators compiled with contraction on: they still fuse)
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                                 // last descriptor whose block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const maf_ema_desc_t e = descs[lo];
    float* __restrict__ dst = static_cast<float*>(e.dst);
    const float* __restrict__ src = static_cast<const float*>(e.src);
    const long long i0 = ((long long)(blockIdx.x - e.block0) * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= e.total && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        f32x4_t a = *reinterpret_cast<const f32x4_t*>(dst + i0);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(src + i0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float t0 = a[q] * d, t1 = b[q] * omd; a[q] = t0 + t1; }
        *reinterpret_cast<f32x4_t*>(dst + i0) = a;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < e.total) { const float t0 = dst[i0 + q] * d, t1 = src[i0 + q] * omd; dst[i0 + q] = t0 + t1; }
}

// GradScaler's inf check (yolov6/core/engine.py:375-391: `self.scaler.step(self.optimizer)` looks for non-finite gradients before the step) over the CONTIGUOUS ranges the
// gradients occupy (the flat buckets of the gradient exchange: a handful of ranges for ~300 tensors) in one launch: *found_inf = 1 if any element is Inf / NaN.  The
// framework's multi-tensor form takes four launches (64 us) for MAF-YOLO-n.
__global__ __launch_bounds__(256) void nonfinite_check_kernel(const maf_range_desc_t* __restrict__ descs, int n, float* __restrict__ found_inf) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                                 // last descriptor whose block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const maf_range_desc_t e = descs[lo];
    const float* __restrict__ src = static_cast<const float*>(e.ptr);
    const long long base = (long long)(blockIdx.x - e.block0) * 4096;           // 4096 elements per block: 4 x 16 bytes per thread
    bool bad = false;
    if (base + 4096 <= e.total && ((uintptr_t)src & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + base + (r * 256 + threadIdx.x) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) bad |= !(fabsf(v[q]) <= 3.402823466e+38f);     // false for Inf and NaN
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + 4096 && i < e.total; i += 256) bad |= !(fabsf(src[i]) <= 3.402823466e+38f);
    }
    if (bad) *found_inf = 1.f;                                        // (every writer stores the same value)
}

// torch.optim.SGD(nesterov, momentum, weight_decay) of the reference's build_optimizer (yolov6/solver/build.py:23-33) for EVERY parameter of every group in ONE launch over a
// descriptor table (as ema_update_kernel), under a GradScaler: found_inf == 1 skips the whole update, grad_scale (when given) un-scales the gradient — and the un-scaled value is
// written back — exactly as the framework's fused implementation does.  The arithmetic follows that implementation operation by operation (the hyper-parameters are DOUBLES
// there: g / scale, g + wd * p, mu * buf + g, g + mu * buf' with the UNROUNDED buf', p - lr * g are formed in double — every multiply-add as ONE fused operation, which is what the
// framework's build contracts them to: located with a dump of its outputs, 47 of 200 000 momentum values differed with separate roundings, none with the fused form — and rounded
// to fp32 once each), so parameters and momentum buffers are bit-identical to it (tests/test_gpu_train.py:test_native_sgd_...).
struct SgdHyper { double lr[MAF_SGD_MAX_GROUPS], wd[MAF_SGD_MAX_GROUPS], mu[MAF_SGD_MAX_GROUPS]; int nesterov[MAF_SGD_MAX_GROUPS]; };

__global__ __launch_bounds__(256) void sgd_update_kernel(const maf_sgd_desc_t* __restrict__ descs, int n, const SgdHyper h, const float* __restrict__ found_inf, const float* __restrict__ grad_scale) {
#pragma clang fp contract(off)
    if (found_inf && *found_inf == 1.f) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                                 // last descriptor whose block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const maf_sgd_desc_t e = descs[lo];
    float* __restrict__ pp = static_cast<float*>(e.param);
    float* __restrict__ gp = static_cast<float*>(e.grad);
    float* __restrict__ bp = static_cast<float*>(e.buf);
    const double lr = h.lr[e.group], wd = h.wd[e.group], mu = h.mu[e.group];
    const bool nest = h.nesterov[e.group] != 0, scaled = grad_scale != nullptr;
    const double scale = scaled ? (double)*grad_scale : 1.0;
    auto one = [&](float p, float g, float b, float& po, float& go, float& bo) {
        if (scaled) { g = (float)((double)g / scale); go = g; }
        if (wd != 0.0) g = (float)__builtin_fma(wd, (double)p, (double)g);
        if (bp) {
            const double nb = __builtin_fma(mu, (double)b, (double)g);  // (dampening 0: the reference's default)
            bo = (float)nb;
            g = nest ? (float)__builtin_fma(mu, nb, (double)g) : (float)nb;
        }
        po = (float)__builtin_fma(-lr, (double)g, (double)p);
    };
    const long long i0 = ((long long)(blockIdx.x - e.block0) * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= e.total && (((uintptr_t)pp | (uintptr_t)gp | (uintptr_t)bp) & 15) == 0) {
        f32x4_t p = *reinterpret_cast<const f32x4_t*>(pp + i0), g = *reinterpret_cast<const f32x4_t*>(gp + i0), b = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (bp) b = *reinterpret_cast<const f32x4_t*>(bp + i0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { float po, go = g[q], bo = b[q]; one(p[q], g[q], b[q], po, go, bo); p[q] = po; g[q] = go; b[q] = bo; }
        *reinterpret_cast<f32x4_t*>(pp + i0) = p;
        if (scaled) *reinterpret_cast<f32x4_t*>(gp + i0) = g;
        if (bp) *reinterpret_cast<f32x4_t*>(bp + i0) = b;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < e.total) {
            float po, go = gp[i0 + q], bo = bp ? bp[i0 + q] : 0.f;
            one(pp[i0 + q], gp[i0 + q], bo, po, go, bo);
            pp[i0 + q] = po;
            if (scaled) gp[i0 + q] = go;
            if (bp) bp[i0 + q] = bo;
        }
}

// Depth-wise weight gradient: dW[c][ky][kx] = sum_{b,y,x} dY[b,y,x,c] * X[b,y+ky-P,x+kx-P,c]   (zero padding).
// Workgroup = one TH x TW tile of one image x one block of CB channels: the X halo tile and the dY tile are staged in
// LDS once; the partial sums reach dW with fp32 atomics (dW is zeroed by the caller).
struct DwWgArgs {
    const void* x; const void* dy; float* dw;
    int B, H, W, C, x_stride, dy_stride, TH, TW, CB, tilesX, tilesY, nCB, replicas, RSEG;
};

// Work item = (16-byte channel group, tap row ky, strip of S = 4 output columns, row segment): for each of its tile rows it loads the
// S dY vectors and the S + K - 1 X vectors of the strip once and performs S x K x 8 multiply-adds (fp16 products into fp32
// accumulators, v_fma_mix_f32): 2S + K - 1 LDS reads per 8*S*K FMAs instead of 2 per 8.  One item per lane; the row segments
// (rows rseg, rseg + RSEG, ...) exist to give EVERY lane an item — groups x K x strips alone are 96 (k = 3) to 288 (k = 9) per
// workgroup, which left 62 % of a 256-lane workgroup idle at k = 3 and made k = 9 take two rounds — and the block size is the item
// count rounded up to whole waves.  With 4 (k >= 7) or 8 channel groups per block the tiles take <= 56 KB of LDS: 2-4 workgroups,
// 3-5 waves per SIMD (one wave per SIMD issues a vector instruction only every ~6 cycles).  A workgroup owns one channel block and
// walks many (image, tile) pairs; its K x N sums per item stay in registers, are summed over strips (lane shuffles) and row segments
// (LDS atomics) at the end and reach dW with one global atomic per (channel, tap).
template <typename T, typename V, int N, int K>
__global__ __launch_bounds__(512) void dw_wgrad_kernel(const DwWgArgs a) {
    constexpr int P = K / 2, S = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    V* xt = reinterpret_cast<V*>(smem_raw);
    const int cb = blockIdx.x % a.nCB, wgi = blockIdx.x / a.nCB, nwg = gridDim.x / a.nCB;
    const int c0 = cb * a.CB;
    const int CGB = min(a.CB, a.C - c0) / N;
    const int PS = a.CB / N + 1;                               // LDS pixel stride in vectors (+16 B pad)
    const int RH = a.TH + K - 1, RW = a.TW + K - 1;
    V* dt = xt + RH * RW * PS;                                 // [TH*TW][PS]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const T* xin = static_cast<const T*>(a.x) + c0;
    const T* dyin = static_cast<const T*>(a.dy) + c0;
    const int nstrip = a.TW / S;
    const int items = CGB * K * nstrip * a.RSEG;               // <= blockDim.x
    const bool live = tid < items;
    const int st = tid % nstrip, r2 = tid / nstrip;            // the strips of one (group, tap row, segment) sit in adjacent lanes
    const int cgi = r2 % CGB, r3 = r2 / CGB;
    const int ky = r3 % K, rseg = r3 / K;
    float acc[K][N];
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int j = 0; j < N; ++j) acc[kx][j] = 0.f;
    const int ntiles = a.B * a.tilesY * a.tilesX;
    for (int tile = wgi; tile < ntiles; tile += nwg) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX;
        const int ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        __syncthreads();                                       // the previous tile has been consumed
        for (int idx = tid; idx < RH * RW * CGB; idx += nthr) {
            const int cg = idx % CGB, p = idx / CGB;
            const int rx = p % RW, ry = p / RW;
            const int iy = y0 - P + ry, ix = x0 - P + rx;
            V v = (V)(T)0;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const V*>(xin + ((size_t)((size_t)b * a.H + iy) * a.W + ix) * a.x_stride + cg * N);
            xt[p * PS + cg] = v;
        }
        for (int idx = tid; idx < a.TH * a.TW * CGB; idx += nthr) {
            const int cg = idx % CGB, p = idx / CGB;
            const int rx = p % a.TW, ry = p / a.TW;
            const int oy = y0 + ry, ox = x0 + rx;
            V v = (V)(T)0;
            if (oy < a.H && ox < a.W)
                v = *reinterpret_cast<const V*>(dyin + ((size_t)((size_t)b * a.H + oy) * a.W + ox) * a.dy_stride + cg * N);
            dt[p * PS + cg] = v;
        }
        __syncthreads();
        if (live) {
            for (int ry = rseg; ry < a.TH; ry += a.RSEG) {
                const V* xr = xt + ((ry + ky) * RW + st * S) * PS + cgi;
                const V* dr = dt + (ry * a.TW + st * S) * PS + cgi;
                V xv[S + K - 1], dv[S];
#pragma unroll
                for (int i = 0; i < S + K - 1; ++i) xv[i] = xr[i * PS];
#pragma unroll
                for (int i = 0; i < S; ++i) dv[i] = dr[i * PS];
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        if constexpr (N == 8) {
                            const u32x4_t xa = __builtin_bit_cast(u32x4_t, xv[i + kx]), da = __builtin_bit_cast(u32x4_t, dv[i]);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[kx][2 * q]) : "v"(xa[q]), "v"(da[q]));
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[kx][2 * q + 1]) : "v"(xa[q]), "v"(da[q]));
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < N; ++j) acc[kx][j] = __builtin_fmaf((float)xv[i + kx][j], (float)dv[i][j], acc[kx][j]);
                        }
                    }
            }
        }
    }
    // strips: lane shuffles (nstrip = 4: xor 1, 2; nstrip = 5: through LDS with the segments); row segments: LDS atomics on a
    // [channel][tap] table in the tile area; then one global atomic per (channel, tap) into replica wgi % replicas — thousands of
    // atomics on the few cache lines of dW serialise, so the workgroups spread over `replicas` copies that the caller adds up
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);           // [CGB * N][K * K]
    const int nred = CGB * N * K * K;
    for (int i = tid; i < nred; i += nthr) red[i] = 0.f;
    __syncthreads();
    const bool pow2 = nstrip == 4;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float v = live ? acc[kx][j] : 0.f;
            if (pow2) {
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
            }
            if (live && (!pow2 || st == 0)) atomicAdd(red + (cgi * N + j) * (K * K) + ky * K + kx, v);
        }
    }
    __syncthreads();
    float* dwr = a.dw + (size_t)(wgi % a.replicas) * a.C * (K * K) + (size_t)c0 * (K * K);
    for (int i = tid; i < nred; i += nthr) atomicAdd(dwr + i, red[i]);
}

template <typename T, typename V, int N>
int launch_dw_wgrad(DwWgArgs& a, int k, hipStream_t s) {
    // Channel groups per block (tools/dw_wgrad_sweep.py; MAF_DWWG = "tile20,gmax,maxthreads,wgcap" overrides): 8 (128-byte pixel rows) on the
    // 160 x 160 maps, which stream; 4 elsewhere — half the LDS, twice the workgroups per CU: 932 -> 727 us over ten shapes of a step —
    // and 2 for the 9 x 9 kernels on 20 x 20 (199 -> 98 us).  20-wide tiles (no waste on 20 / 40 / 80-wide maps) measured slower: five
    // strips per row are not a power of two for the lane shuffles.
    int tile20 = 0, gmax = (long long)a.H * a.W >= 160 * 160 ? 8 : (k == 9 && a.H * a.W <= 400) ? 2 : 4, maxthr = 256, wgcap = 1024, th = 0, tw = 0, abudget = 1500000;
    if (const char* e = getenv("MAF_DWWG")) sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &tile20, &gmax, &maxthr, &wgcap, &th, &tw, &abudget);
    a.TH = (tile20 && a.H % 10 == 0 && a.H <= 40) ? 10 : min(8, a.H);
    a.TW = (tile20 && a.W % 20 == 0) ? 20 : 16;
    if (th > 0 && tw > 0 && tw % 4 == 0) { a.TH = min(th, a.H); a.TW = tw; }      // sweep override (tools/dw_wgrad_sweep.py)
    const int groups = a.C / N, nblk = maf_cdiv(groups, gmax);
    const int cgb = maf_cdiv(groups, nblk);
    a.CB = cgb * N;                                                     // balanced channel blocks (72 channels: 40 + 32)
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    const int base = cgb * k * (a.TW / 4);                              // items without row segments
    a.RSEG = maxthr / base > a.TH ? a.TH : maxthr / base > 0 ? maxthr / base : 1;
    MAF_REQUIRE(base * a.RSEG <= 512, "dw_wgrad: work items exceed the workgroup");
    const int nthr = (base * a.RSEG + 63) / 64 * 64;
    size_t lds = ((size_t)(a.TH + k - 1) * (a.TW + k - 1) + (size_t)a.TH * a.TW) * (cgb + 1) * 16;
    const size_t red = (size_t)a.CB * k * k * 4;
    if (lds < red) lds = red;
    int per = a.B * a.tilesY * a.tilesX;                                // workgroups per channel block
    const int cap = maf_cdiv(wgcap, a.nCB);
    if (per > cap) per = cap;
    const int budget = (int)((long long)abudget / ((long long)a.C * k * k));      // every workgroup ends with C*k*k/nCB atomics: keep their total ~1.5 M
    if (per > budget) per = budget > 8 ? budget : 8;
    const dim3 g(per * a.nCB), b(nthr);
    switch (k) {
        case 1: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 1>), g, b, lds, s, a); break;      // the 1 x 1 branch of a DilatedReparamBlock (a per-channel scale): dW[c] = sum dy x
        case 3: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 3>), g, b, lds, s, a); break;
        case 5: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 5>), g, b, lds, s, a); break;
        case 7: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 7>), g, b, lds, s, a); break;
        case 9: hipLaunchKernelGGL((dw_wgrad_kernel<T, V, N, 9>), g, b, lds, s, a); break;
        default: maf_set_error("dw_wgrad: k must be 1, 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
    return maf_check_hip(hipGetLastError(), "dw_wgrad launch");
}

// The 3 x 3 (+ 3 x 3) + 1 x 1 branches of a train-form DilatedReparamBlock that share ONE input (kernel sets 3,3,1 and 5,3,1: common.py:2997-3008) in one launch.
// As separate launches every branch stages the X halo tile again: 80 x 80 x 192, batch 32: k3 61.7 us + k1 50.6 us for 3 x 78.6 MB of X + dY reads each way; the 1 x 1
// branch — dW1[c] = sum dY1 * X, a per-channel scale — needs nothing but the centre tap of the tile the 3 x 3 item already holds.  Work items as in dw_wgrad_kernel with
// K = 3 (channel group, tap row, strip, row segment); an item multiplies its X strip with the strip of dYa (and dYb); the items of the CENTRE tap row also take dY1.
struct DwWg31Args {
    const void* x; const void* dya; const void* dyb; const void* dy1;
    float* dwa; float* dwb; float* dw1;
    int B, H, W, C, x_stride, dya_stride, dyb_stride, dy1_stride, TH, TW, CB, tilesX, tilesY, nCB, replicas, RSEG;
};

template <typename T, typename V, int N, bool HASB>
__global__ __launch_bounds__(512) void dw_wgrad31_kernel(const DwWg31Args a) {
    constexpr int K = 3, P = 1, S = 4, NT = HASB ? 3 : 2;        // dY tiles in LDS: a, [b,] 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    V* xt = reinterpret_cast<V*>(smem_raw);
    const int cb = blockIdx.x % a.nCB, wgi = blockIdx.x / a.nCB, nwg = gridDim.x / a.nCB;
    const int c0 = cb * a.CB;
    const int CGB = min(a.CB, a.C - c0) / N;
    const int PS = a.CB / N + 1;
    const int RH = a.TH + K - 1, RW = a.TW + K - 1, TP = a.TH * a.TW;
    V* dt = xt + RH * RW * PS;                                 // [NT][TH*TW][PS]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const T* xin = static_cast<const T*>(a.x) + c0;
    const T* dsrc[3] = {static_cast<const T*>(a.dya) + c0, HASB ? static_cast<const T*>(a.dyb) + c0 : static_cast<const T*>(a.dy1) + c0, static_cast<const T*>(a.dy1) + c0};
    const int dstr[3] = {a.dya_stride, HASB ? a.dyb_stride : a.dy1_stride, a.dy1_stride};
    const int nstrip = a.TW / S;
    const int items = CGB * K * nstrip * a.RSEG;
    const bool live = tid < items;
    const int st = tid % nstrip, r2 = tid / nstrip;
    const int cgi = r2 % CGB, r3 = r2 / CGB;
    const int ky = r3 % K, rseg = r3 / K;
    const bool mid = live && ky == P;
    float acca[K][N], accb[HASB ? K : 1][N], acc1[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        acc1[j] = 0.f;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) { acca[kx][j] = 0.f; if constexpr (HASB) accb[kx][j] = 0.f; }
    }
    auto fma8 = [](float* acc, const V& xv, const V& dv) {
        if constexpr (N == 8) {
            const u32x4_t xa = __builtin_bit_cast(u32x4_t, xv), da = __builtin_bit_cast(u32x4_t, dv);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q]) : "v"(xa[q]), "v"(da[q]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q + 1]) : "v"(xa[q]), "v"(da[q]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] = __builtin_fmaf((float)xv[j], (float)dv[j], acc[j]);
        }
    };
    const int ntiles = a.B * a.tilesY * a.tilesX;
    for (int tile = wgi; tile < ntiles; tile += nwg) {
        const int tx = tile % a.tilesX, t2 = tile / a.tilesX;
        const int ty = t2 % a.tilesY, b = t2 / a.tilesY;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        __syncthreads();
        for (int idx = tid; idx < RH * RW * CGB; idx += nthr) {
            const int cg = idx % CGB, p = idx / CGB;
            const int rx = p % RW, ry = p / RW;
            const int iy = y0 - P + ry, ix = x0 - P + rx;
            V v = (V)(T)0;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const V*>(xin + ((size_t)((size_t)b * a.H + iy) * a.W + ix) * a.x_stride + cg * N);
            xt[p * PS + cg] = v;
        }
        for (int idx = tid; idx < NT * TP * CGB; idx += nthr) {
            const int cg = idx % CGB, q = idx / CGB;
            const int p = q % TP, t = q / TP;
            const int rx = p % a.TW, ry = p / a.TW;
            const int oy = y0 + ry, ox = x0 + rx;
            V v = (V)(T)0;
            if (oy < a.H && ox < a.W)
                v = *reinterpret_cast<const V*>(dsrc[t] + ((size_t)((size_t)b * a.H + oy) * a.W + ox) * dstr[t] + cg * N);
            dt[(t * TP + p) * PS + cg] = v;
        }
        __syncthreads();
        if (live) {
            for (int ry = rseg; ry < a.TH; ry += a.RSEG) {
                const V* xr = xt + ((ry + ky) * RW + st * S) * PS + cgi;
                const V* dr = dt + (ry * a.TW + st * S) * PS + cgi;
                V xv[S + K - 1], dv[S];
#pragma unroll
                for (int i = 0; i < S + K - 1; ++i) xv[i] = xr[i * PS];
#pragma unroll
                for (int i = 0; i < S; ++i) dv[i] = dr[i * PS];
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) fma8(acca[kx], xv[i + kx], dv[i]);
                if constexpr (HASB) {
#pragma unroll
                    for (int i = 0; i < S; ++i) dv[i] = dr[(TP + i) * PS];
#pragma unroll
                    for (int i = 0; i < S; ++i)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) fma8(accb[kx], xv[i + kx], dv[i]);
                }
                if (mid) {
#pragma unroll
                    for (int i = 0; i < S; ++i) dv[i] = dr[((NT - 1) * TP + i) * PS];
#pragma unroll
                    for (int i = 0; i < S; ++i) fma8(acc1, xv[i + P], dv[i]);
                }
            }
        }
    }
    // reduction as in dw_wgrad_kernel: strips by lane shuffles, row segments by LDS atomics, one global atomic per (channel, tap) and branch
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);           // [CGB * N][9] a, [CGB * N][9] b, [CGB * N] 1
    const int n9 = CGB * N * K * K, nred = (HASB ? 2 : 1) * n9 + CGB * N;
    for (int i = tid; i < nred; i += nthr) red[i] = 0.f;
    __syncthreads();
    const bool pow2 = nstrip == 4;
    auto fold = [&](float v, int slot, bool on) {
        if (!on) v = 0.f;
        if (pow2) {
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
        }
        if (on && (!pow2 || st == 0)) atomicAdd(red + slot, v);
    };
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            fold(acca[kx][j], (cgi * N + j) * (K * K) + ky * K + kx, live);
            if constexpr (HASB) fold(accb[kx][j], n9 + (cgi * N + j) * (K * K) + ky * K + kx, live);
        }
        fold(acc1[j], (HASB ? 2 : 1) * n9 + cgi * N + j, mid);
    }
    __syncthreads();
    const size_t rep = (size_t)(wgi % a.replicas);
    float* da = a.dwa + rep * a.C * (K * K) + (size_t)c0 * (K * K);
    for (int i = tid; i < n9; i += nthr) atomicAdd(da + i, red[i]);
    if constexpr (HASB) {
        float* db = a.dwb + rep * a.C * (K * K) + (size_t)c0 * (K * K);
        for (int i = tid; i < n9; i += nthr) atomicAdd(db + i, red[n9 + i]);
    }
    float* d1 = a.dw1 + rep * a.C + c0;
    for (int i = tid; i < CGB * N; i += nthr) atomicAdd(d1 + i, red[(HASB ? 2 : 1) * n9 + i]);
}

template <typename T, typename V, int N>
int launch_dw_wgrad31(DwWg31Args& a, hipStream_t s) {
    // geometry of launch_dw_wgrad for k = 3; channel groups per block / threads from tools/dw_wgrad31_sweep.py (batch 32, us, separate launches -> merged):
    //   160 x 160 x 72 (3+3+1)  278 -> 243 with <= 8 groups (5 + 4: 4 or 3 leave 48-byte pixel rows: 316);  80 x 80 x 192  113 -> 95 with 8 groups on 512 threads (4: 115);
    //   80 x 80 x 144  105 -> 84 and 80 x 80 x 128  80 -> 60 with 4 groups (8: 84 / 89)
    constexpr int k = 3;
    const bool big = (long long)a.H * a.W >= 160 * 160, wide = a.C >= 160;
    int gmax = big || wide ? 8 : 4, maxthr = wide && !big ? 512 : 256, wgcap = 1024, abudget = 1500000;
    if (const char* e = getenv("MAF_DWWG31")) sscanf(e, "%d,%d,%d,%d", &gmax, &maxthr, &wgcap, &abudget);
    a.TH = min(8, a.H); a.TW = 16;
    const int groups = a.C / N, nblk = maf_cdiv(groups, gmax);
    const int cgb = maf_cdiv(groups, nblk);
    a.CB = cgb * N;
    a.tilesX = maf_cdiv(a.W, a.TW); a.tilesY = maf_cdiv(a.H, a.TH); a.nCB = maf_cdiv(a.C, a.CB);
    const int base = cgb * k * (a.TW / 4);
    a.RSEG = maxthr / base > a.TH ? a.TH : maxthr / base > 0 ? maxthr / base : 1;
    MAF_REQUIRE(base * a.RSEG <= 512, "dw_wgrad31: work items exceed the workgroup");
    const int nthr = (base * a.RSEG + 63) / 64 * 64;
    const int nt = a.dyb ? 3 : 2;
    size_t lds = ((size_t)(a.TH + k - 1) * (a.TW + k - 1) + (size_t)nt * a.TH * a.TW) * (cgb + 1) * 16;
    const size_t red = (size_t)a.CB * (2 * k * k + 1) * 4;
    if (lds < red) lds = red;
    int per = a.B * a.tilesY * a.tilesX;
    const int cap = maf_cdiv(wgcap, a.nCB);
    if (per > cap) per = cap;
    const int budget = (int)((long long)abudget / ((long long)a.C * (nt - 1) * k * k));
    if (per > budget) per = budget > 8 ? budget : 8;
    const dim3 g(per * a.nCB), b(nthr);
    static bool attr = false;                                            // (8 groups x three dY tiles: 81 KB of LDS)
    if (!attr) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad31_kernel<T, V, N, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(dw_wgrad31)");
        if (!rc) rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad31_kernel<T, V, N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(dw_wgrad31)");
        if (rc) return rc;
        attr = true;
    }
    if (a.dyb) hipLaunchKernelGGL((dw_wgrad31_kernel<T, V, N, true>), g, b, lds, s, a);
    else hipLaunchKernelGGL((dw_wgrad31_kernel<T, V, N, false>), g, b, lds, s, a);
    return maf_check_hip(hipGetLastError(), "dw_wgrad31 launch");
}

// dst[b, 2y, 2x, :] += src[b, y, x, :] (NHWC, 16-byte channel chunks): the data gradient of a stride-2 1x1 conv (RepVGGBlock.rbr_1x1, common.py:203) added onto the
// 3x3 branch's data gradient of the same input — instead of a zero-filled full-size tensor, a strided copy and a full-size add.
template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void add_sub2_kernel(const T* __restrict__ src, int src_stride, T* __restrict__ dst, int dst_stride, int Ho, int Wo, int CG, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i % CG);
    long long p = i / CG;
    const int x = (int)(p % Wo); p /= Wo;
    const int y = (int)(p % Ho);
    const long long b = p / Ho;
    const V a = *reinterpret_cast<const V*>(src + ((b * Ho + y) * Wo + x) * src_stride + cg * N);
    V* q = reinterpret_cast<V*>(dst + ((b * 2 * Ho + 2 * y) * (2ll * Wo) + 2 * x) * dst_stride + cg * N);
    V d = *q;
#pragma unroll
    for (int j = 0; j < N; ++j) d[j] = (T)((float)d[j] + (float)a[j]);
    *q = d;
}

// Per-channel sum over the pixels of an NHWC tensor into fp32 (the bias gradient of a conv with bias: cls_pred / reg_pred, common.py:1304-1305): thread = (channel,
// pixel slab) with the channel fastest (coalesced for any channel count), slabs of a block reduced in LDS, one atomic per channel and block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int stride, long long M, int C, float* __restrict__ out) {
    __shared__ float s_part[256];
    const int per = 256 / C > 0 ? 256 / C : 1;                       // pixel slabs per block (C <= 256)
    const int c = threadIdx.x % C, slab = threadIdx.x / C;
    float acc = 0.f;
    if (slab < per) {
        // eight independent 2-byte loads in flight per lane (one at a time: 42 us for the 32 x 80 x 80 x 80 class logits, 0.77 TB/s)
        const long long step = (long long)gridDim.x * per;
        long long m = (long long)blockIdx.x * per + slab;
        for (; m + 7 * step < M; m += 8 * step) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(m + u * step) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (float)v[u];
        }
        for (; m < M; m += step) acc += (float)x[m * stride + c];
    }
    s_part[threadIdx.x] = slab < per ? acc : 0.f;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int j = 0; j < per; ++j) t += s_part[j * C + threadIdx.x];
        atomicAdd(out + threadIdx.x, t);
    }
}

// Gradient fold (maf_grad_fold): the weight-gradient kernels leave dW in THEIR layout — tap-major [taps][Cout_p][Cin_p] with the channel
// counts padded to what the kernels read (8-channel chunks) — and the optimizer wants the parameter's [Cout][Cin][taps]: one small launch adds
// (or writes) the valid part into the gradient slice, on the stream the weight gradient ran on.
__global__ void grad_fold_kernel(const float* __restrict__ src, int taps, int co_p, int ci_p, float* __restrict__ dst, int cout, int cin, int accumulate, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = i % taps, r = i / taps, ci = r % cin, co = r / cin;
    const float v = src[((size_t)t * co_p + co) * ci_p + ci];
    dst[i] = accumulate ? dst[i] + v : v;
}

// dst = [dst +] sum_i src[i] on NHWC views (maf_nhwc_sum): thread = one 16-byte channel group of one pixel, the group index fastest (a pixel's groups are one contiguous run)
struct SumArgs {
    const void* src[4]; int ss[4]; void* dst; int ds, n, acc, CG; long long total;
};
template <typename T, typename V, int N>
__global__ __launch_bounds__(256) void nhwc_sum_kernel(const SumArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int cg = (int)(i % a.CG);
    const long long m = i / a.CG;
    V* q = reinterpret_cast<V*>(static_cast<T*>(a.dst) + m * a.ds + cg * N);
    if (a.n == 1 && !a.acc) { *q = *reinterpret_cast<const V*>(static_cast<const T*>(a.src[0]) + m * a.ss[0] + cg * N); return; }
    float s[N];
    V v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < a.n) v[k] = *reinterpret_cast<const V*>(static_cast<const T*>(a.src[k]) + m * a.ss[k] + cg * N);
    if (a.acc) { const V d = *q;
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = (float)d[j];
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < a.n) {
#pragma unroll
            for (int j = 0; j < N; ++j) s[j] += (float)v[k][j];
        }
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (T)s[j];
    *q = o;
}

hipEvent_t g_fork_ev[16] = {};

}  // namespace

extern "C" int maf_nhwc_sum(const void* const* src, const int32_t* src_stride, int32_t n, void* dst, int32_t dst_stride, int64_t M, int32_t C, int32_t dtype,
                            int32_t accumulate, maf_stream_t stream) {
    MAF_REQUIRE(src && src_stride && dst && n >= 1 && n <= 4 && M > 0 && C > 0, "nhwc_sum: bad arguments (1..4 sources)");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "nhwc_sum: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && dst_stride % N == 0, "nhwc_sum: C and strides must be multiples of the 16-byte channel group");
    SumArgs a = {};
    for (int i = 0; i < n; ++i) {
        MAF_REQUIRE(src[i] && src_stride[i] % N == 0, "nhwc_sum: null source / stride not a multiple of the 16-byte channel group");
        a.src[i] = src[i]; a.ss[i] = src_stride[i];
    }
    a.dst = dst; a.ds = dst_stride; a.n = n; a.acc = accumulate ? 1 : 0; a.CG = C / N; a.total = (long long)M * a.CG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((a.total + 255) / 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((nhwc_sum_kernel<half_t, half8_t, 8>), g, b, 0, s, a);
    else hipLaunchKernelGGL((nhwc_sum_kernel<float, f32x4_t, 4>), g, b, 0, s, a);
    return maf_check_hip(hipGetLastError(), "nhwc_sum launch");
}

// fork / join of the weight-gradient stream by raw handles: one reusable event per device (a wait captures the record that precedes it)
extern "C" int maf_stream_fork(maf_stream_t main, maf_stream_t side) {
    if (main == side) return 0;
    int dev = 0;
    if (int rc = maf_check_hip(hipGetDevice(&dev), "stream_fork: hipGetDevice")) return rc;
    MAF_REQUIRE(dev >= 0 && dev < 16, "stream_fork: device index out of range");
    if (!g_fork_ev[dev])
        if (int rc = maf_check_hip(hipEventCreateWithFlags(&g_fork_ev[dev], hipEventDisableTiming), "stream_fork: hipEventCreate")) return rc;
    if (int rc = maf_check_hip(hipEventRecord(g_fork_ev[dev], static_cast<hipStream_t>(main)), "stream_fork: hipEventRecord")) return rc;
    return maf_check_hip(hipStreamWaitEvent(static_cast<hipStream_t>(side), g_fork_ev[dev], 0), "stream_fork: hipStreamWaitEvent");
}

extern "C" int maf_stream_join(maf_stream_t main, maf_stream_t side) { return maf_stream_fork(side, main); }

extern "C" int maf_grad_fold(const float* src, int32_t taps, int32_t Cout_p, int32_t Cin_p, float* dst, int32_t Cout, int32_t Cin, int32_t accumulate,
                             maf_stream_t stream) {
    MAF_REQUIRE(src && dst && taps > 0 && Cout > 0 && Cin > 0 && Cout <= Cout_p && Cin <= Cin_p, "grad_fold: bad arguments");
    const long long total = (long long)taps * Cout * Cin;
    MAF_REQUIRE(total < (1ll << 31), "grad_fold: tensor too large");
    hipLaunchKernelGGL(grad_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src, taps, Cout_p, Cin_p, dst, Cout, Cin,
                       accumulate, (int)total);
    return maf_check_hip(hipGetLastError(), "grad_fold launch");
}

extern "C" int maf_add_sub2(const void* src, int32_t src_stride, void* dst, int32_t dst_stride, int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, maf_stream_t stream) {
    MAF_REQUIRE(src && dst && B > 0 && Ho > 0 && Wo > 0 && C > 0, "add_sub2: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "add_sub2: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && src_stride % N == 0 && dst_stride % N == 0, "add_sub2: C and strides must be multiples of the 16-byte channel group");
    const long long total = (long long)B * Ho * Wo * (C / N);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (dtype == MAF_F16) hipLaunchKernelGGL((add_sub2_kernel<half_t, half8_t, 8>), g, b, 0, s, static_cast<const half_t*>(src), src_stride, static_cast<half_t*>(dst), dst_stride, Ho, Wo, C / N, total);
    else hipLaunchKernelGGL((add_sub2_kernel<float, f32x4_t, 4>), g, b, 0, s, static_cast<const float*>(src), src_stride, static_cast<float*>(dst), dst_stride, Ho, Wo, C / N, total);
    return maf_check_hip(hipGetLastError(), "add_sub2 launch");
}

extern "C" int maf_colsum(const void* x, int32_t x_stride, int64_t M, int32_t C, int32_t dtype, float* out, maf_stream_t stream) {
    MAF_REQUIRE(x && out && M > 0 && C > 0 && C <= 256 && x_stride >= C, "colsum: bad arguments (C <= 256)");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "colsum: dtype must be f16/f32");
    const int per = 256 / C > 0 ? 256 / C : 1;
    long long nb = (M + per * 64 - 1) / (per * 64);                   // >= 64 pixels per thread
    nb = nb < 1 ? 1 : nb > 1024 ? 1024 : nb;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) hipLaunchKernelGGL((colsum_kernel<half_t>), dim3((unsigned)nb), dim3(256), 0, s, static_cast<const half_t*>(x), x_stride, (long long)M, C, out);
    else hipLaunchKernelGGL((colsum_kernel<float>), dim3((unsigned)nb), dim3(256), 0, s, static_cast<const float*>(x), x_stride, (long long)M, C, out);
    return maf_check_hip(hipGetLastError(), "colsum launch");
}

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

extern "C" int maf_pack_batch(const maf_pack_desc_t* descs_dev, int32_t n, int32_t nblocks, maf_stream_t stream) {
    MAF_REQUIRE(descs_dev && n > 0 && nblocks > 0, "pack_batch: bad arguments");
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), descs_dev, n);
    return maf_check_hip(hipGetLastError(), "pack_batch launch");
}

extern "C" int32_t maf_pack_desc_size(void) { return (int32_t)sizeof(maf_pack_desc_t); }

extern "C" int maf_ema_update(const maf_ema_desc_t* descs_dev, int32_t n, int32_t nblocks, float decay, float one_minus_decay, maf_stream_t stream) {
    MAF_REQUIRE(descs_dev && n > 0 && nblocks > 0, "ema_update: bad arguments");
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), descs_dev, n, decay, one_minus_decay);
    return maf_check_hip(hipGetLastError(), "ema_update launch");
}

extern "C" int32_t maf_ema_desc_size(void) { return (int32_t)sizeof(maf_ema_desc_t); }

extern "C" int maf_sgd_update(const maf_sgd_desc_t* descs_dev, int32_t n, int32_t nblocks, int32_t ngroups, const double* lr, const double* weight_decay, const double* momentum,
                              const int32_t* nesterov, const float* found_inf, const float* grad_scale, maf_stream_t stream) {
    MAF_REQUIRE(descs_dev && n > 0 && nblocks > 0 && lr && weight_decay && momentum && nesterov, "sgd_update: bad arguments");
    MAF_REQUIRE(ngroups >= 1 && ngroups <= MAF_SGD_MAX_GROUPS, "sgd_update: 1..8 parameter groups");
    SgdHyper h;
    for (int i = 0; i < MAF_SGD_MAX_GROUPS; ++i) {
        const int j = i < ngroups ? i : 0;
        h.lr[i] = lr[j]; h.wd[i] = weight_decay[j]; h.mu[i] = momentum[j]; h.nesterov[i] = nesterov[j];
    }
    hipLaunchKernelGGL(sgd_update_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), descs_dev, n, h, found_inf, grad_scale);
    return maf_check_hip(hipGetLastError(), "sgd_update launch");
}

extern "C" int32_t maf_sgd_desc_size(void) { return (int32_t)sizeof(maf_sgd_desc_t); }

extern "C" int maf_nonfinite_check(const maf_range_desc_t* descs_dev, int32_t n, int32_t nblocks, float* found_inf, maf_stream_t stream) {
    MAF_REQUIRE(descs_dev && n > 0 && nblocks > 0 && found_inf, "nonfinite_check: bad arguments");
    hipLaunchKernelGGL(nonfinite_check_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), descs_dev, n, found_inf);
    return maf_check_hip(hipGetLastError(), "nonfinite_check launch");
}

extern "C" int32_t maf_range_desc_size(void) { return (int32_t)sizeof(maf_range_desc_t); }

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
                            int32_t k, int32_t dtype, float* dw, int32_t replicas, maf_stream_t stream) {
    MAF_REQUIRE(replicas >= 1 && replicas <= 64, "dw_wgrad: replicas (copies of dW the workgroups spread their atomics over) must be 1..64");
    MAF_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && C > 0, "dw_wgrad: bad arguments");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "dw_wgrad: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && x_stride % N == 0 && dy_stride % N == 0, "dw_wgrad: C and strides must be multiples of the 16-byte channel group");
    DwWgArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.B = B; a.H = H; a.W = W; a.C = C; a.x_stride = x_stride; a.dy_stride = dy_stride; a.replicas = replicas;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16 && k >= 3) {
        static const int mf = getenv("MAF_DWWG_MFMA") ? atoi(getenv("MAF_DWWG_MFMA")) : -1;      // 1: wherever it fits, 0: never, default: the maps it wins on
        // tools/dw_wgrad_ab.py, batch 32: 20 x 20 k 9 / 7 / 5 / 3: 101 / 57 / 45 / 27 -> 22 / 20 / 16 / 15 us (576 channels), 40 x 40 k 7 / 5: 56 / 41 -> 28 / 27 (192), 80 x 80 k 5: 97 -> 71 (192);
        // k = 3 on the 80 x 80 maps and on 40 x 40 x 288 stays with the vector kernel (62 / 32 us against 62 - 70 / 35)
        if (mf == 1 || (mf < 0 && W <= 96 && (k >= 5 || H * W <= 400 || (H * W <= 1600 && C <= 192)))) {
            const int rc = maf_dw_wgrad_mfma(x, x_stride, dy, dy_stride, B, H, W, C, k, dw, replicas, s);
            if (rc != MAF_E_UNSUPPORTED) return rc;
        }
    }
    if (dtype == MAF_F16) return launch_dw_wgrad<half_t, half8_t, 8>(a, k, s);
    return launch_dw_wgrad<float, f32x4_t, 4>(a, k, s);
}

extern "C" int maf_dw_wgrad31(const void* x, int32_t x_stride, const void* dya, int32_t dya_stride, const void* dyb, int32_t dyb_stride, const void* dy1, int32_t dy1_stride,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, float* dwa, float* dwb, float* dw1, int32_t replicas, maf_stream_t stream) {
    MAF_REQUIRE(replicas >= 1 && replicas <= 64, "dw_wgrad31: replicas must be 1..64");
    MAF_REQUIRE(x && dya && dy1 && dwa && dw1 && (!dyb == !dwb) && B > 0 && H > 0 && W > 0 && C > 0, "dw_wgrad31: bad arguments (x, dya / dwa, dy1 / dw1 required; dyb and dwb together)");
    MAF_REQUIRE(dtype == MAF_F16 || dtype == MAF_F32, "dw_wgrad31: dtype must be f16/f32");
    const int N = dtype == MAF_F16 ? 8 : 4;
    MAF_REQUIRE(C % N == 0 && x_stride % N == 0 && dya_stride % N == 0 && dy1_stride % N == 0 && (!dyb || dyb_stride % N == 0),
                "dw_wgrad31: C and strides must be multiples of the 16-byte channel group");
    DwWg31Args a;
    a.x = x; a.dya = dya; a.dyb = dyb; a.dy1 = dy1; a.dwa = dwa; a.dwb = dwb; a.dw1 = dw1; a.B = B; a.H = H; a.W = W; a.C = C;
    a.x_stride = x_stride; a.dya_stride = dya_stride; a.dyb_stride = dyb_stride; a.dy1_stride = dy1_stride; a.replicas = replicas;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MAF_F16) return launch_dw_wgrad31<half_t, half8_t, 8>(a, s);
    return launch_dw_wgrad31<float, f32x4_t, 4>(a, s);
}
