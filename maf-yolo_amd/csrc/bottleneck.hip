// Fused DepthBottleneckUni (deploy form): 1x1 (c -> 3c) + SiLU -> depth-wise k x k + SiLU -> 1x1 (3c -> c) + SiLU in ONE
// kernel; the two 3c-channel intermediates never leave the CU and all three convolutions run on the matrix cores.
//
// Replaces, for one bottleneck, Conv.forward_fuse x2 + the merged UniRepLKNetBlock + DepthBottleneckUni.act of
// yolov6/layers/common.py:918-927 (conv1 :905, conv2 :908, act :909, one_conv :910) — three launches that wrote and
// re-read the 3c-wide tensors twice: per bottleneck the HBM traffic drops from (2c + 12c) to 2c channels per pixel.
//
// One workgroup (4 waves) = one 16 x 16 output tile of one image; the mid channels are processed in blocks of 32:
//
//   A. T1 = SiLU(X * W1[:, block] + b1) on the (16+k-1)^2 halo tile     MFMA 16x16x32; A fragments are 16-byte global loads
//      (exact zeros outside the image = the depth-wise conv's padding); T1 goes to LDS *planar* ([channel][row][x], 8-byte
//      stores of the 4 consecutive pixels an accumulator lane owns).
//   B. T2 = SiLU(DW_k(T1) + bdw)                                          MFMA 16x16x32 with a block-diagonal Toeplitz operand:
//      one instruction convolves FOUR channels along x for one tap row ky:  A[(G,r)][(g,j)] = (G==g) * w[ch_G][ky][j - r]
//      (4 outputs x 8-wide input window per channel), B[(g,j)][n] = T1[ch_g][row n + ky][x window j] — a 16-byte LDS read per
//      lane; k instructions (2k for k > 5: two windows) accumulate a [4 ch][4 x][16 rows] block.  ~15 % of the MACs are useful
//      and it still is 5-9x the VALU rate (v_fma_mix: 16 MAC/clk/SIMD; this: 80-144).
//      After the 8 channel sets of a block, lane (G, n) holds channels 8G..8G+7 of pixels (row n, x = 4q + 0..3) — which IS
//      the A fragment of the second 1x1 (lane (g, i): 8 consecutive k of row i), so T2 never touches LDS.
//   C. acc2 += T2 * W2[block, :]                                          MFMA, accumulators stay in registers across blocks
//   epilogue: out = SiLU(acc2 + b2), NHWC into the concat slice.
//
// The 1x1 on the halo is recomputed ((16+k-1)^2/256 times: its K is only c).  The remaining VALU work is the two SiLUs.
// fp16 storage, fp32 accumulation (same arithmetic as the three separate kernels up to the fp16 rounding of T1/T2, which
// those kernels apply when they store them; the depth-wise weights are fp16 there too).
//
// Fused tail (nc = C3 > 0, template C3T = C3 / 16, NX): the 1x1 conv that closes the surrounding RepHDW block (common.py:944-946:
// conv2(cat(x1, x2, .., y)) + SiLU) is applied to the tile before anything is stored — the bottleneck's output y is never written and
// the closing conv never launched.  Phase C then runs TRANSPOSED (operands swapped: accumulator rows = channels, columns = pixel rows), so
// that after bias + SiLU a lane holds 8-channel runs of ONE pixel: the B fragments of the tail GEMM, no LDS round trip; the other concat
// sources are read as B fragments straight from global (16-byte loads: the centre of the halo tile phase A has just pulled through L2) and
// the tail's weight fragments (A operand: rows = output channels, permuted so that the four lane groups of a store instruction cover one
// contiguous 64-byte run of a pixel) arrive by DMA in the LDS the block loop no longer needs.
#include "maf_common.h"
#include <type_traits>

namespace {

struct BnArgs {
    const half_t* x; half_t* out;
    const unsigned char* par;   // [nMB] block records: Toeplitz [8][K][PARTS][16] half8 | W1 fragments [2][S1][64] half8 | W2 fragments [CT2][64] half8 | b1 [32] f32 | bdw [32] f32
    const float* b2;
    int B, H, W, Cin, Cout, nMB, x_stride, x_coff, out_stride, out_coff;
    int tilesX, tilesY, nwg;
    int ko;
    // fused tail: sources of the closing 1x1 in concat order (xs[NX - 1] = this bottleneck's own input), its record (pack.py:pack_bottleneck_tail)
    const half_t* xs[3];
    int xs_stride[3], xs_coff[3];
    const unsigned char* w3;
    int C3, nx;
    unsigned* ctr;              // MAF_BN_PERSIST builds: [8] next dynamic tile of an XCD's segment, [8] = workgroups that have left (the last one resets the set)
    int ntiles;
    unsigned long long* prof;   // MAF_BN_PROFILE builds only (op->aux[3]): cycles per phase summed over all waves, see tools/bn_profile.py
};

#ifdef MAF_BN_PROFILE
#define BN_KO(bit) ((a.ko >> (bit)) & 1)     // knock-out switches of the profiling build (op->aux[2]): results are wrong, times tell what a piece costs
#else
#define BN_KO(bit) 0
#endif
#ifdef MAF_BN_STAMPS
// s_memtime is a scalar-memory op (counts on lgkmcnt): every stamp drains the counter, so stamps sit only at phase boundaries
#define BN_STAMP(i) do { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); prof_acc[i] += (uint32_t)(now_ - prof_t); prof_t = now_; } while (0)
#else
#define BN_STAMP(i) do { } while (0)
#endif

typedef half_t half4v_t __attribute__((ext_vector_type(4)));

template <int N, int I = 0, typename F>
__device__ __forceinline__ void maf_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        maf_static_for<N, I + 1>(f);
    }
}

// LDS reads as inline assembly (the caller counts them and waits with bn_wait_lgkm): see phase B.
template <int OFF> __device__ __forceinline__ void bn_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void bn_ds_read_b64(u32x2_t& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void bn_wait_lgkm(u32x4_t& a, u32x2_t& b, u32x2_t& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }
template <int N> __device__ __forceinline__ void bn_wait_lgkm1(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void bn_wait_lgkm2(u32x4_t& a, u32x4_t& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// SiLU of a value that arrives pre-multiplied by log2(e) (pack_bottleneck folds the factor into W1 / b1 / bdw and its inverse into W2):
// returns log2(e) * silu(x) for the argument log2(e) * x — v_exp_f32 is a base-2 exponential, so no multiplication in front of it.
__device__ __forceinline__ float bn_silu2(float v) { return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v)); }
// (two at a time with the add and the multiplication as v_pk_add_f32 / v_pk_mul_f32 — 6 issue slots for two values instead of 8 — made the
// kernel TWICE as slow: 68 -> 150 us; packed fp32 arithmetic is no full-rate path on this part.  Scalar it stays.)
// MAF_BN_PKSILU (round 5, an experiment build: `make var VAR=pk VARFLAGS=-DMAF_BN_PKSILU`): the two SiLUs of a pixel pair in PACKED fp16 — the operand is
// rounded to fp16 first (what the unfused path does when it stores the conv output, common.py:44-50 under --half), then v_exp_f16 x2, ONE v_pk_add_f16,
// v_rcp_f16 x2, ONE v_pk_mul_f16: 7 instructions per pair instead of 9, within 2 fp16 ulp of the fp32 form.  Measured (tools/ab_lib.sh, one box, two runs each):
// <5,2,4,8,2> 83.8 -> 81.8 us, <3,1,2,3,2> 92.1 -> 90.2, <5,2,4,6,2> 70.4 -> 67.8; value 25 780 -> 25 960 images/s.  The 19 kernel / model tests of the fused
// bottleneck pass with it, but the extra rounding noise pushed the 32 x 640^2 cross-plan comparison (test_fused_stem_matches_unfused[u8-n], boxes within 0.3 px
// over 268 800 rows) over its bar: 2 us per launch do not buy a wider parity bar, so the product keeps the fp32 form.
// The compiler splits such code into scalar converts and v_pack_b32_f16 (22 instructions for four values); written with SDWA sub-dword selects:
// (two pairs interleaved: a sub-dword write needs a wait state in front of its reader, which the other pair's instruction fills)
__device__ __forceinline__ void bn_silu2_pk4(float v0, float v1, float v2, float v3, half2_t& o01, half2_t& o23) {
    const half2_t xa = {(half_t)v0, (half_t)v1}, xb = {(half_t)v2, (half_t)v3};      // v_cvt_pk_f16_f32
    const uint32_t ua = __builtin_bit_cast(uint32_t, xa), ub = __builtin_bit_cast(uint32_t, xb);
    uint32_t ea, eb, ra, rb;
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(ea) : "v"(ua));
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(eb) : "v"(ub));
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(ea) : "v"(ua));
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(eb) : "v"(ub));
    const half2_t one = {(half_t)1.f, (half_t)1.f};
    const uint32_t da = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, ea) + one), db = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, eb) + one);   // v_pk_add_f16
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(ra) : "v"(da));
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(rb) : "v"(db));
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(ra) : "v"(da));
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(rb) : "v"(db));
    o01 = xa * __builtin_bit_cast(half2_t, ra);                                       // v_pk_mul_f16
    o23 = xb * __builtin_bit_cast(half2_t, rb);
}
// MAF_BN_PERSIST (round 5, an experiment build: `make var VAR=ps VARFLAGS=-DMAF_BN_PERSIST`; VERDICT r4 #2a): resident workgroups (occupancy x 256 CUs) that pull
// tiles dynamically — first tile = blockIdx's own, then one atomicAdd per tile on the counter of the workgroup's XCD segment (64 counter sets handed out
// round-robin so launches in flight on different streams never share one; the workgroup that leaves last re-arms its set); the next tile's index is requested
// when a tile starts, the zero table is written once.  No spills (the lane index is made opaque per tile, else the tile loop keeps 53 hoisted registers), +120
// instructions.  Measured: SLOWER on every instantiation — <5,2,4,8,2> 83.8 -> 103.5 us, <3,1,2,3,2> 92.1 -> 125.5, <5,2,4,6,2> 70.4 -> 89.3 (round 4's static
// stride: 94.6 -> 101.6).  800 tiles on 512 slots take two tile times either way; a likely reason is that the resident form loses the hardware dispatcher's overlap of a
// finishing workgroup's epilogue (stores in flight) with its successor's prologue on the freed half of the CU, while a resident workgroup serialises the two behind
// its own barriers.  Dropped; the switch stays for the measurement.
#ifdef MAF_BN_PERSIST
constexpr bool BN_PERSIST = true;
#else
constexpr bool BN_PERSIST = false;
#endif
#ifdef MAF_BN_PKSILU           // make var VAR=pk VARFLAGS=-DMAF_BN_PKSILU
constexpr bool BN_PK = true;
#else
constexpr bool BN_PK = false;
#endif

template <int K, int S1, int CT2>
struct BnCfg {
    static constexpr int P = K / 2, PARTS = K > 5 ? 2 : 1;
    static constexpr int RH = 16 + K - 1;                          // halo rows
    static constexpr int RWC = (16 + K - 1 + 3) & ~3;              // halo columns computed (whole 4-pixel runs)
    static constexpr int NHP = RH * RWC, NPT = (NHP + 15) / 16;    // halo pixels, 16-pixel m-tiles
    static constexpr int MT = (NPT + 3) / 4;                       // m-tiles per wave
    static constexpr int RWP = 24;                                 // T1 row stride in halfs: 48 B — the 16 rows of a B fragment tile the 256-B bank row
    static constexpr int PSB = ((RH * RWP * 2 - 8 + 255) / 256) * 256 + 8;     // channel plane stride in bytes, = 8 (mod 256): see the bank notes below
    static constexpr int PS = PSB / 2;
    static constexpr int NTOE = 8 * K * PARTS * 16;                // Toeplitz entries (16 B each) per block
    // block record = part A (operands of the first 1x1: W1 | b1) + part B (depth-wise + second 1x1: Toeplitz | W2 | bdw)
    static constexpr int OFF_W1 = 0, OFF_B1 = 2 * S1 * 1024, REC_A = OFF_B1 + 128;
    static constexpr int OFF_TOE = REC_A, OFF_W2 = OFF_TOE + NTOE * 16, OFF_BD = OFF_W2 + CT2 * 1024;
    static constexpr int REC = OFF_BD + 128;                       // bytes per block record
    static constexpr int REC_B = REC - REC_A;
    // ZT: the 48 lanes outside the block diagonal of the Toeplitz operand read a ZERO table of the same size instead of masking what they
    // read (4 v_and per MFMA in a VALU-bound kernel); where the extra NTOE * 16 bytes would cost a workgroup per CU the mask stays
    // OCC3 (-DMAF_BN_OCC3, an experiment): three workgroups per CU for the c <= 64 instantiations too — <= 168 registers and <= 53 KB of LDS, i.e.
    // part B single-buffered again and no zero table: one more wave per SIMD to issue vector instructions from
#ifdef MAF_BN_OCC3
    static constexpr bool OCC3 = true;
#else
    static constexpr bool OCC3 = false;
#endif
    static constexpr bool ZT = CT2 == 4 && K <= 5 && !OCC3;
    static constexpr bool DB = !OCC3;                              // part B double-buffered
    // LDS: T1 planes | part A of the current block | part B twice (the next block's part B arrives while this block's is being read) | zero table
    static constexpr size_t LDS = (size_t)32 * PSB + (size_t)REC + (DB ? (size_t)REC_B : 0) + (ZT ? (size_t)NTOE * 16 : 0);
    static constexpr int ZOFF = REC + (DB ? REC_B : 0);            // offset of the zero table behind the record buffers
};

template <int K, int S1, int CT2, int C3T = 0, int NX = 0>
__global__ __launch_bounds__(256, ((CT2 == 2 && K <= 5) || (BnCfg<K, S1, CT2>::OCC3 && K <= 5)) ? 3 : 2) void bottleneck_kernel(const BnArgs a) {
    typedef BnCfg<K, S1, CT2> Cf;
    constexpr int P = Cf::P, PARTS = Cf::PARTS, RWC = Cf::RWC, NHP = Cf::NHP, NPT = Cf::NPT, MT = Cf::MT, RWP = Cf::RWP, PS = Cf::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* T1 = reinterpret_cast<half_t*>(smem_raw);                        // [32][PS]
    unsigned char* rec = smem_raw + 32 * Cf::PSB;                            // [REC_A][REC_B][REC_B]: part A of the current block (refilled as soon as phase A is over), part B of this and of the next block

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    int lid;
    __shared__ int s_next;
    const int my_xcd = blockIdx.x & 7;
    const int seg_q = a.ntiles >> 3, seg_r = a.ntiles & 7;              // XCD x owns tiles [seg_start(x), + seg_len(x)) of the XCD-contiguous order
    auto seg_start = [&](int x) { return x < seg_r ? x * (seg_q + 1) : seg_r * (seg_q + 1) + (x - seg_r) * seg_q; };
    auto seg_stat = [&](int x) { return ((int)gridDim.x >> 3) + (x < ((int)gridDim.x & 7) ? 1 : 0); };      // tiles of the segment taken statically (one per resident workgroup)
    auto fetch_tile = [&]() -> int {                                    // thread 0: the next tile of this XCD's segment (segments differ by one tile at most: no stealing)
        const int j = seg_stat(my_xcd) + (int)atomicAdd(a.ctr + my_xcd, 1u);
        return j < seg_q + (my_xcd < seg_r ? 1 : 0) ? seg_start(my_xcd) + j : -1;
    };
    {   // XCD-contiguous tile order: neighbouring tiles (shared halos) run on the same XCD's L2
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int q = a.ntiles >> 3, r = a.ntiles & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    bool first_tile = true;
    for (;;) {                                                          // (one pass unless BN_PERSIST)
    // (the lane index is made opaque per tile: otherwise every lane-derived address and mask of the body is hoisted out of the tile loop and kept in
    // registers across it — 53 more than the kernel has)
    int tid_l = threadIdx.x;
    if constexpr (BN_PERSIST) asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, wave = tid >> 6, lane = tid & 63, g = lane >> 4, p = lane & 15;
    int next_tile = -1;
    if (BN_PERSIST && tid == 0) next_tile = fetch_tile();
    const int tx = lid % a.tilesX;
    const int tt = lid / a.tilesX;
    const int ty = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty * 16, x0 = tx * 16;
    const half_t* xin = a.x + (size_t)b * a.H * a.W * a.x_stride + a.x_coff;

    // the wave's share of the halo pixels (m-tiles wave, wave + 4, ...): the same global addresses for every block.
    // Loads are unconditional: pixels outside the image are clamped to a valid pixel (their T1 is forced to zero when it is
    // written) and channel groups beyond Cin to group 0 (their W1 rows are zero), so no lane ever reads outside the tensor.
    const int cg0 = g * 8 < a.Cin ? g * 8 : 0;
    const int cg1 = 32 + g * 8 < a.Cin ? 32 + g * 8 : 0;
    uint32_t xoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wave + 4 * i) * 16 + p;
        const int hr = m / RWC, hc = m - hr * RWC;
        const int iy = min(max(y0 - P + hr, 0), a.H - 1), ix = min(max(x0 - P + hc, 0), a.W - 1);
        xoff[i] = (uint32_t)(iy * a.W + ix) * a.x_stride;
    }
    int ko_block = 0;
#ifndef MAF_BN_XD
#define MAF_BN_XD 3         // 2 -> 3: -1 ... -3 us per launch (the MFMAs of m-tile i + 1 waited for loads issued one m-tile section earlier); 4 spills the 168-register forms, 5 gains nothing
#endif
    constexpr int XD = Cf::OCC3 ? 1 : (MAF_BN_XD < MT ? MAF_BN_XD : MT - 1);   // activation fragments are loaded XD m-tiles ahead of their MFMAs
    half8_t af[XD + 1][S1];
    auto load_x = [&](auto idx) {
        constexpr int i = decltype(idx)::value;
        if constexpr (i < MT) {
            if (BN_KO(0) && ko_block > 0) return;            // KO 0: activations are read for the first mid block only
            af[i % (XD + 1)][0] = *reinterpret_cast<const half8_t*>(xin + xoff[i] + cg0);
            if constexpr (S1 > 1) af[i % (XD + 1)][1] = *reinterpret_cast<const half8_t*>(xin + xoff[i] + cg1);
        }
    };
    auto load_x_head = [&]() {
        maf_static_for<XD>([&](auto idx) { load_x(idx); });
    };
    // zero mask of the T1 values this lane writes: accumulator lane (g, p) of m-tile t owns halo pixels t*16 + 4g .. +3
    // (one row, 4 consecutive columns); bit r set = pixel r is inside the image.  interior tiles: all ones.
    const bool interior = y0 - P >= 0 && x0 - P >= 0 && y0 - P + Cf::RH <= a.H && x0 - P + RWC <= a.W;
    // Per m-tile, for all blocks: where its 4 pixels go in a T1 plane (-1: past the halo tile) and which of them lie inside the image (the others
    // are the depth-wise conv's zero padding).  A wave issues one vector instruction every ~6 cycles whatever the pipe could take
    // (tools/valu_probe.py), so every instruction that can leave the block loop does.
    constexpr int MTH = Cf::OCC3 ? 1 : MT;                  // (the three-workgroup experiment has no registers for them: computed per block)
    uint32_t tmlo[MTH], tmhi[MTH];
    int tdst[MTH];
#pragma unroll
    for (int i = 0; i < (Cf::OCC3 ? 0 : MT); ++i) {
        const int m0 = (wave + 4 * i) * 16 + g * 4;
        const int hr = m0 / RWC, hc0 = m0 - hr * RWC;
        tmlo[i] = tmhi[i] = 0xffffffffu;
        if (!interior) {
            const int iy = y0 - P + hr, ixb = x0 - P + hc0;
            const bool rowok = (unsigned)iy < (unsigned)a.H;
            const uint32_t k0 = (rowok && (unsigned)(ixb + 0) < (unsigned)a.W) ? 0x0000ffffu : 0u, k1 = (rowok && (unsigned)(ixb + 1) < (unsigned)a.W) ? 0xffff0000u : 0u;
            const uint32_t k2 = (rowok && (unsigned)(ixb + 2) < (unsigned)a.W) ? 0x0000ffffu : 0u, k3 = (rowok && (unsigned)(ixb + 3) < (unsigned)a.W) ? 0xffff0000u : 0u;
            tmlo[i] = k0 | k1; tmhi[i] = k2 | k3;
        }
        tdst[i] = m0 < NHP ? p * PS + hr * RWP + hc0 : -1;
    }
    // The next block's record travels global -> LDS by DMA (global_load_lds: no registers, no ds_write; every wave moves 1 KiB
    // per instruction to a wave-uniform LDS base + lane * 16).  The barrier that publishes it also waits for it (vmcnt).
    auto dma_part = [&](int mb, int off, int bytes, int dst) {
        const unsigned char* src = a.par + (size_t)mb * Cf::REC + off;
        const int nvec = bytes >> 4;
        for (int v0 = wave * 64; v0 < nvec; v0 += 256) {
            if (v0 + lane < nvec)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)(v0 + lane) * 16),
                                                 (void __attribute__((address_space(3)))*)(rec + dst + v0 * 16), 16, 0, 0);
        }
    };

    f32x4_t acc2[4][CT2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) acc2[i][ct] = (f32x4_t)0.f;

    const uint32_t toe_mask = (p >> 2) == g ? 0xffffffffu : 0u;      // block-diagonal Toeplitz: lane (g, (G, r)) is non-zero iff G == g
    const int q4 = wave * 4;                                // this wave's 4 output columns
    int hi_off = 4;
    asm volatile("" : "+v"(hi_off));                        // opaque: the two 8-byte halves of a window stay two ds_read_b64 (2 LDS cycles each; ds_read2_b64 costs 8)

#ifdef MAF_BN_STAMPS
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t prof_t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t) :: "memory");
#endif
    dma_part(0, 0, Cf::REC, 0);
    load_x_head();
    if constexpr (Cf::ZT) {
        if (first_tile) {
            u32x4_t* z = reinterpret_cast<u32x4_t*>(rec + Cf::ZOFF);
            for (int i = tid; i < Cf::NTOE; i += 256) z[i] = (u32x4_t){0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    BN_STAMP(0);                                            // prologue

    for (int mb = 0; mb < a.nMB; ++mb) {
        ko_block = mb;
        if (!Cf::DB && mb > 0) dma_part(mb, Cf::REC_A, Cf::REC_B, Cf::REC_A);      // single buffer: free since the barrier that ended the previous block
        const unsigned char* recB = rec + (Cf::DB ? (mb & 1) * Cf::REC_B : 0);    // this block's part B (offsets OFF_TOE / OFF_W2 / OFF_BD count from the record start)
        // ---- A. T1 = SiLU(X * W1[:, block] + b1) on the halo tile
        if (!BN_KO(8)) {                                    // KO 8: no phase A
            half8_t w1f[2][S1];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int ks = 0; ks < S1; ++ks) w1f[ct][ks] = reinterpret_cast<const half8_t*>(rec + Cf::OFF_W1)[(ct * S1 + ks) * 64 + lane];
            const float b1v0 = reinterpret_cast<const float*>(rec + Cf::OFF_B1)[p], b1v1 = reinterpret_cast<const float*>(rec + Cf::OFF_B1)[16 + p];
            // the MFMAs of m-tile i + 1 are issued BEFORE the SiLU / store work of m-tile i: a wave issues in order, so written tile by tile the
            // matrix pipe idles during the ~50 vector instructions of a tile and the vector pipe during the MFMAs and their result latency
            f32x4_t accn[2];
            auto mma_tile = [&](auto idx) {
                constexpr int i = decltype(idx)::value;
                if constexpr (i < MT) {
                    accn[0] = (f32x4_t)b1v0; accn[1] = (f32x4_t)b1v1;                // bias as the accumulators' initial value: no add in the epilogue
#pragma unroll
                    for (int ks = 0; ks < S1; ++ks)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const half8_t wv = Cf::OCC3 ? reinterpret_cast<const half8_t*>(rec + Cf::OFF_W1)[(ct * S1 + ks) * 64 + lane] : w1f[ct][ks];   // OCC3: no registers to keep W1 in
                            accn[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i % (XD + 1)][ks], wv, accn[ct], 0, 0, 0);
                        }
                }
            };
            if constexpr (!Cf::OCC3) mma_tile(std::integral_constant<int, 0>{});
            maf_static_for<MT>([&](auto idx) {
                constexpr int i = decltype(idx)::value;
                const int t = wave + 4 * i;
                load_x(std::integral_constant<int, i + XD>{});
                if constexpr (Cf::OCC3) mma_tile(idx);
                f32x4_t acc1[2] = {accn[0], accn[1]};
                if constexpr (!Cf::OCC3) mma_tile(std::integral_constant<int, i + 1>{});
                __builtin_amdgcn_sched_barrier(0);                                       // keep the next tile's MFMAs in front of this tile's vector work
                uint32_t mlo, mhi;                                     // keep masks and LDS address of this m-tile: the same for every block, computed once
                int doff;
                if constexpr (!Cf::OCC3) {
                    mlo = tmlo[i]; mhi = tmhi[i]; doff = tdst[i];
                } else {
                    const int m0 = t * 16 + g * 4;
                    const int hr = m0 / RWC, hc0 = m0 - hr * RWC;
                    mlo = mhi = 0xffffffffu;
                    if (!interior) {
                        const int iy = y0 - P + hr, ixb = x0 - P + hc0;
                        const bool rowok = (unsigned)iy < (unsigned)a.H;
                        mlo = ((rowok && (unsigned)(ixb + 0) < (unsigned)a.W) ? 0x0000ffffu : 0u) | ((rowok && (unsigned)(ixb + 1) < (unsigned)a.W) ? 0xffff0000u : 0u);
                        mhi = ((rowok && (unsigned)(ixb + 2) < (unsigned)a.W) ? 0x0000ffffu : 0u) | ((rowok && (unsigned)(ixb + 3) < (unsigned)a.W) ? 0xffff0000u : 0u);
                    }
                    doff = m0 < NHP ? p * PS + hr * RWP + hc0 : -1;
                }
                half_t* dst = T1 + doff;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    half2_t h01, h23;
                    if (BN_KO(1)) {                                                        // KO 1: no SiLU in phase A (a branch: as a select both sides were computed and the switch measured nothing)
                        h01 = half2_t{(half_t)acc1[ct][0], (half_t)acc1[ct][1]}; h23 = half2_t{(half_t)acc1[ct][2], (half_t)acc1[ct][3]};
                    } else if constexpr (BN_PK) {
                        bn_silu2_pk4(acc1[ct][0], acc1[ct][1], acc1[ct][2], acc1[ct][3], h01, h23);
                    } else {
                        h01 = half2_t{(half_t)bn_silu2(acc1[ct][0]), (half_t)bn_silu2(acc1[ct][1])};
                        h23 = half2_t{(half_t)bn_silu2(acc1[ct][2]), (half_t)bn_silu2(acc1[ct][3])};
                    }
                    const u32x2_t w = {__builtin_bit_cast(uint32_t, h01) & mlo, __builtin_bit_cast(uint32_t, h23) & mhi};
                    if (doff >= 0 && !BN_KO(4)) *reinterpret_cast<u32x2_t*>(dst + (size_t)(16 * ct) * PS) = w;                                        // KO 4: no T1 stores
                }
            });
        }
        BN_STAMP(1);                                        // phase A
        if (!BN_KO(7)) __syncthreads();                     // KO 7: no barriers inside the block loop
        BN_STAMP(2);                                        // barrier after A
        if (mb + 1 < a.nMB) {
            // phase A is over: its operands can be replaced; the other part-B buffer was last read before the barrier that ended the
            // previous block.  Both DMAs are issued BEFORE the activation loads: vmcnt counts in order, so the wait in front of the next
            // phase A's first MFMA (which the compiler has to write as vmcnt(0)) then waits for loads that are a whole phase B + C old,
            // not for a DMA issued a few instructions earlier (the single-buffer version stalled there once per block).
            if (!BN_KO(5)) {                                 // KO 5: the first block's operands for every block
            dma_part(mb + 1, 0, Cf::REC_A, 0);
            if (Cf::DB) dma_part(mb + 1, Cf::REC_A, Cf::REC_B, Cf::REC_A + ((mb + 1) & 1) * Cf::REC_B);
            }
            load_x_head();                                  // the next phase A's first activations: in flight during phase B
        }

        // ---- B. depth-wise k x k on the matrix cores: 8 channel sets s (channels 8g + s), k tap rows, PARTS windows
        f32x4_t dacc[8];                                    // initial value = the depth-wise bias of channel 4s + g (no add in phase C)
        {
            const f32x4_t bd0 = reinterpret_cast<const f32x4_t*>(recB + Cf::OFF_BD)[g * 2], bd1 = reinterpret_cast<const f32x4_t*>(recB + Cf::OFF_BD)[g * 2 + 1];
#pragma unroll
            for (int s = 0; s < 4; ++s) { dacc[s] = (f32x4_t)bd0[s]; dacc[4 + s] = (f32x4_t)bd1[s]; }
        }
        const half8_t* toe = reinterpret_cast<const half8_t*>(recB + Cf::OFF_TOE) + p;
        // lane (g, n = p): plane 4s + g, row n (+ky), window at column 4q, read as two 8-byte halves.  Banks: the 16 rows of a
        // lane group are 48 B apart (all 16-byte slots of the 256-B bank row once) and the planes of g and g+1 are 8 B (mod 256)
        // apart, so the 32 lanes of a ds_read_b64 group cover the 64 banks exactly once; the 8-byte stores of phase A
        // (16 consecutive planes per group, 8 B apart mod 128) are conflict-free for the same reason.
        const half_t* t1l = T1 + (size_t)g * PS + p * RWP + q4;
        const half_t* t1h = t1l + hi_off;
        // The K * PARTS * 8 steps (tap row, window, channel set) are ONE straight line of code, software-pipelined by hand: the three LDS
        // reads of step t + BD are issued before the MFMA of step t, so an MFMA never waits for a read it has just issued (the
        // loop form did: `s_waitcnt lgkmcnt(0)` in front of every MFMA, ~150 cycles per step for a 17-cycle instruction, because the
        // Toeplitz read sat under a lane predicate and every step became its own basic block).  All 64 lanes read a Toeplitz entry
        // (lane p's: in range for every lane) and the 48 lanes outside the block diagonal AND it to zero.
        // The reads are inline assembly with hand-counted `s_waitcnt lgkmcnt(n)` (LDS returns in order: when step t is consumed the
        // 3 * min(BD, steps left) reads issued after its own may still be in flight); written as plain loads the compiler waits with
        // lgkmcnt(0) — a full LDS round trip — on every (BD+1)-th step.
        // B128 (MAF_BN_B128): one ds_read_b128 per window although its address is only 8-byte aligned for the odd waves.  It works (gfx950 runs
        // the LDS in unaligned-access mode) and it is a disaster: 68 -> 150 us — a misaligned 16-byte read is not one access.  Two ds_read_b64 stay.
#ifdef MAF_BN_B128
        constexpr bool B128 = true;
#else
        constexpr bool B128 = false;
#endif
        constexpr int NSTEP = K * PARTS * 8, BD = (CT2 == 2 || Cf::OCC3) ? 2 : 3;    // read-ahead depth (the c <= 32 variants run 3 workgroups per CU on 168 registers)
        u32x4_t avr[BD + 1];
        u32x4_t bwin[BD + 1];                                  // B128: the 16-byte window in one (8-byte aligned) read; else two 8-byte halves
        u32x2_t blo[BD + 1], bhi[BD + 1];
        const half8_t* toe_src = toe;
        if constexpr (Cf::ZT) { if (!toe_mask) toe_src = reinterpret_cast<const half8_t*>(rec + Cf::ZOFF) + p; }
        const uint32_t a_toe = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)toe_src;
        const uint32_t a_t1l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)t1l;
        const uint32_t a_t1h = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)t1h;
        auto ld_step = [&](auto idx) {
            constexpr int t = decltype(idx)::value;
            if constexpr (t < NSTEP) {
                constexpr int s = t % 8, part = (t / 8) % PARTS, ky = t / (8 * PARTS);
                constexpr int ot = ((s * K + ky) * PARTS + part) * 256, o = (s * 4 * PS + ky * RWP + part * 4) * 2;
                static_assert(ot < 65536 && o < 65536, "ds offset field");
                bn_ds_read_b128<ot>(avr[t % (BD + 1)], a_toe);
                if constexpr (B128) {
                    bn_ds_read_b128<o>(bwin[t % (BD + 1)], a_t1l);
                } else {
                    bn_ds_read_b64<o>(blo[t % (BD + 1)], a_t1l);
                    bn_ds_read_b64<o>(bhi[t % (BD + 1)], a_t1h);
                }
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the compiler's own reads (biases) are complete: the counter now counts only the reads below
        if (!BN_KO(2)) {                                      // KO 2: no phase B
        maf_static_for<BD>([&](auto idx) { ld_step(idx); });
        maf_static_for<NSTEP>([&](auto idx) {
            constexpr int t = decltype(idx)::value, s = t % 8, sl = t % (BD + 1);
            ld_step(std::integral_constant<int, t + BD>{});
            constexpr int ahead = (NSTEP - 1 - t) < BD ? (NSTEP - 1 - t) : BD;      // steps whose reads were issued after step t's
            u32x4_t bw;
            if constexpr (B128) {
                bn_wait_lgkm2<2 * ahead>(avr[sl], bwin[sl]);
                bw = bwin[sl];
            } else {
                bn_wait_lgkm<3 * ahead>(avr[sl], blo[sl], bhi[sl]);
                bw = (u32x4_t){blo[sl][0], blo[sl][1], bhi[sl][0], bhi[sl][1]};
            }
            const u32x4_t am = Cf::ZT ? avr[sl] : (avr[sl] & toe_mask);
            dacc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, am), __builtin_bit_cast(half8_t, bw), dacc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);              // keep the issue order as written
        });
        }
        BN_STAMP(3);                                        // phase B
        // ---- C. lane (g, n) now owns channels 4s + g (s = 0..7: k index 8g + s of the packed W2) of pixels (row n, x = 4q + r): the A fragments of the second 1x1
        {
            half8_t w2f[CT2];
#pragma unroll
            for (int ct = 0; ct < CT2; ++ct) w2f[ct] = reinterpret_cast<const half8_t*>(recB + Cf::OFF_W2)[ct * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                half8_t t2;
#pragma unroll
                for (int s = 0; s < 8; ++s) t2[s] = (half_t)dacc[s][r];
                if (!BN_KO(3)) {                                                           // KO 3: no SiLU in phase C
                    if constexpr (BN_PK) {
#pragma unroll
                        for (int s = 0; s < 8; s += 4) {
                            half2_t ha, hb;
                            bn_silu2_pk4(dacc[s][r], dacc[s + 1][r], dacc[s + 2][r], dacc[s + 3][r], ha, hb);
                            t2[s] = ha[0]; t2[s + 1] = ha[1]; t2[s + 2] = hb[0]; t2[s + 3] = hb[1];
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < 8; ++s) t2[s] = (half_t)bn_silu2(dacc[s][r]);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < CT2; ++ct) if (!BN_KO(9)) {   // KO 9: no second 1x1
                    if constexpr (C3T > 0) acc2[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ct], t2, acc2[r][ct], 0, 0, 0);   // transposed: rows = channels (4g + rr) CT2 + ct, columns = pixel rows
                    else acc2[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t2, w2f[ct], acc2[r][ct], 0, 0, 0);
                }
            }
        }
        BN_STAMP(4);                                        // phase C
        if (mb + 1 < a.nMB && !BN_KO(7)) {
            __syncthreads();                                // phase B has finished reading T1 and part B; part A of the next block is visible
        }
        BN_STAMP(5);                                        // barrier after C
    }

    if constexpr (C3T > 0) {
    // ---- fused tail: out = SiLU(W3 * cat(xs[0..NX-1], y) + b3), y = SiLU(acc2 + b2) from the (transposed) accumulators.
    // Lane (g, p) of acc2[r][ct], register rr: channel (4g + rr) CT2 + ct of pixel (row p, x = 4q + r) — 4 CT2 consecutive channels from 4g CT2:
    // k-step j of the tail's y part takes channels 4g CT2 + 8j .. + 7 (pack.py:pack_bottleneck_tail orders W3's rows to match).
    constexpr int YS = CT2 / 2, NV = 4 * C3T, PSZ = (NV % 8 == 0) ? 8 : 4, NPC = NV / PSZ, NXS = NX * S1, NFR = (NXS + YS) * C3T;
    static_assert(NFR * 1024 + 64 * C3T <= (int)Cf::LDS, "the tail's record fits the LDS of the block loop");
    static_assert((NFR - 1) * 1024 < 65536, "ds offset field");
    __syncthreads();                                               // every wave has finished reading T1 and the block records
    {
        constexpr int nvec = (NFR * 1024 + 64 * C3T) >> 4;         // A fragments [step][t3][64 lanes] 16 B | b3 [4 g][C3T][4 rr] f32
        for (int v0 = wave * 64; v0 < nvec; v0 += 256)
            if (v0 + lane < nvec)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.w3 + (size_t)(v0 + lane) * 16),
                                                 (void __attribute__((address_space(3)))*)(smem_raw + v0 * 16), 16, 0, 0);
    }
    // B fragments of the other concat sources: lane (g, p) = pixel (row p, x = 4q + r), channels 32 ks + 8g .. + 7 (groups past the source's
    // c channels read group 0 against zero weight rows); item i = (half h, source s) is loaded while item i - 1 is multiplied
    const int oy = y0 + p, oyc = min(oy, a.H - 1);
    const size_t rowpix = ((size_t)b * a.H + oyc) * a.W;
    int cgk[S1];
#pragma unroll
    for (int ks = 0; ks < S1; ++ks) cgk[ks] = 32 * ks + 8 * g < a.Cin ? 32 * ks + 8 * g : 0;
    half8_t xf[2][2][S1];
    auto load_item = [&](auto idx) {
        constexpr int i = decltype(idx)::value;
        if constexpr (i < 2 * NX) {
            constexpr int h = i / NX, sx = i % NX;
#pragma unroll
            for (int rl = 0; rl < 2; ++rl) {
                const int oxc = min(x0 + q4 + 2 * h + rl, a.W - 1);
                const half_t* px = a.xs[sx] + (rowpix + oxc) * a.xs_stride[sx] + a.xs_coff[sx];
#pragma unroll
                for (int ks = 0; ks < S1; ++ks) xf[i & 1][rl][ks] = *reinterpret_cast<const half8_t*>(px + cgk[ks]);
            }
        }
    };
    load_item(std::integral_constant<int, 0>{});
    // (bias + SiLU of the bottleneck's own output while the record and the first fragments travel)
    half8_t yf[4][YS];
    {
        float b2v[4 * CT2];
#pragma unroll
        for (int i = 0; i < 4 * CT2; ++i) b2v[i] = a.b2[g * 4 * CT2 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < YS; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = 8 * j + q, rr = idx / CT2, ct = idx % CT2;
                    yf[r][j][q] = (half_t)maf_act<MAF_ACT_SILU>(acc2[r][ct][rr] + b2v[idx]);
                }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                               // the record has landed
    const uint32_t a_w3 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(smem_raw + lane * 16);
    const f32x4_t* b3l = reinterpret_cast<const f32x4_t*>(smem_raw + NFR * 1024) + g * C3T;
    half_t* obase = a.out + a.out_coff + g * PSZ;
    maf_static_for<2>([&](auto hidx) {
        constexpr int h = decltype(hidx)::value;
        f32x4_t acc3[2][C3T];                                      // [pixel x = 4q + 2h + rl][tile t3]: rows 4g + rr = output value rr C3T + t3 of this lane
#pragma unroll
        for (int t3 = 0; t3 < C3T; ++t3) { acc3[0][t3] = b3l[t3]; acc3[1][t3] = acc3[0][t3]; }
        // one group of k-steps (a source, or y): NKS * C3T (k-step, tile) steps in one straight line, the A fragment of step t + RD read before the
        // two MFMAs of step t (hand-counted waits, as in phase B)
        auto group = [&](auto base_c, auto nks_c, auto&& getB) {
            constexpr int FB = decltype(base_c)::value, NKS = decltype(nks_c)::value, NST = NKS * C3T, RD = NST > 4 ? 4 : NST - 1;
            u32x4_t wr[RD + 1];
            auto ld = [&](auto idx) {
                constexpr int t = decltype(idx)::value;
                if constexpr (t < NST) bn_ds_read_b128<(FB + t) * 1024>(wr[t % (RD + 1)], a_w3);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            maf_static_for<RD>([&](auto idx) { ld(idx); });
            maf_static_for<NST>([&](auto idx) {
                constexpr int t = decltype(idx)::value, ks = t / C3T, t3 = t % C3T, sl = t % (RD + 1);
                ld(std::integral_constant<int, t + RD>{});
                constexpr int ahead = (NST - 1 - t) < RD ? (NST - 1 - t) : RD;
                bn_wait_lgkm1<ahead>(wr[sl]);
                const half8_t wv = __builtin_bit_cast(half8_t, wr[sl]);
                acc3[0][t3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, getB(0, ks), acc3[0][t3], 0, 0, 0);
                acc3[1][t3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, getB(1, ks), acc3[1][t3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        maf_static_for<NX>([&](auto sidx) {
            constexpr int sx = decltype(sidx)::value, i = h * NX + sx;
            load_item(std::integral_constant<int, i + 1>{});
            group(std::integral_constant<int, sx * S1 * C3T>{}, std::integral_constant<int, S1>{}, [&](int rl, int ks) { return xf[i & 1][rl][ks]; });
        });
        group(std::integral_constant<int, NXS * C3T>{}, std::integral_constant<int, YS>{}, [&](int rl, int j) { return yf[2 * h + rl][j]; });
        if (!BN_KO(6)) {
#pragma unroll
        for (int rl = 0; rl < 2; ++rl) {
            const int ox = x0 + q4 + 2 * h + rl;
            if (oy < a.H && ox < a.W) {
                half_t* op = obase + (((size_t)b * a.H + oy) * a.W + ox) * a.out_stride;
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    uint32_t w[PSZ / 2];
#pragma unroll
                    for (int e = 0; e < PSZ / 2; ++e) {
                        const int v0 = pc * PSZ + 2 * e, v1 = v0 + 1;
                        const half2_t hv = {(half_t)maf_act<MAF_ACT_SILU>(acc3[rl][v0 % C3T][v0 / C3T]), (half_t)maf_act<MAF_ACT_SILU>(acc3[rl][v1 % C3T][v1 / C3T])};
                        w[e] = __builtin_bit_cast(uint32_t, hv);
                    }
                    if constexpr (PSZ == 8) *reinterpret_cast<u32x4_t*>(op + pc * 32) = (u32x4_t){w[0], w[1], w[2 % (PSZ / 2)], w[3 % (PSZ / 2)]};
                    else *reinterpret_cast<u32x2_t*>(op + pc * 16) = (u32x2_t){w[0], w[1]};
                }
            }
        }
        }
    });
    } else {
    // ---- epilogue: out = SiLU(acc2 + b2); accumulator lane (g, p), register rr: pixel (row 4g + rr, x = 4q + r), channels p*CT2 ..
    // Stored straight from the accumulators a lane would issue 16 stores of CT2 halfs (8 bytes): store-issue bound (10 us of the kernel).
    // Instead every wave stages its 16 x 4 pixels in its own slice of the (now free) T1 area, pixel-major, and writes 16-byte pieces:
    // 8 lanes cover the whole channel run of a pixel, 4x fewer store instructions.
    float bias2[CT2];
#pragma unroll
    for (int ct = 0; ct < CT2; ++ct) bias2[ct] = a.b2[p * CT2 + ct];
    constexpr int CO = 16 * CT2;                                   // staged channels per pixel (>= Cout)
    static_assert(4 * 64 * CO * 2 <= 32 * Cf::PSB, "epilogue staging fits the T1 area");
    __syncthreads();                                               // every wave has finished reading T1 (phase B of the last block)
    half_t* stg = T1 + wave * (64 * CO);                           // [64 pixels = (row 0..15) x (col 0..3)][CO]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            uint32_t w[CT2 / 2];
#pragma unroll
            for (int c2 = 0; c2 < CT2 / 2; ++c2) {
                const half2_t h = {(half_t)maf_act<MAF_ACT_SILU>(acc2[r][2 * c2][rr] + bias2[2 * c2]),
                                   (half_t)maf_act<MAF_ACT_SILU>(acc2[r][2 * c2 + 1][rr] + bias2[2 * c2 + 1])};
                w[c2] = __builtin_bit_cast(uint32_t, h);
            }
            half_t* sp = stg + ((g * 4 + rr) * 4 + r) * CO + p * CT2;
            if (CT2 == 4) *reinterpret_cast<u32x2_t*>(sp) = (u32x2_t){w[0], w[1 % (CT2 / 2)]};
            else *reinterpret_cast<uint32_t*>(sp) = w[0];
        }
    }
    // a wave reads back only what it wrote itself: LDS operations of one wave complete in order, no barrier
    const int cpp = a.Cout >> 3;                                   // 16-byte pieces per pixel (Cout is a multiple of 8)
    half_t* obase = a.out + (size_t)b * a.H * a.W * a.out_stride + a.out_coff;
    if (!BN_KO(6)) {                                               // KO 6: no output stores
    for (int q = lane; q < 64 * cpp; q += 64) {
        const int px = q / cpp, part = q - px * cpp;
        const int oy = y0 + (px >> 2), ox = x0 + q4 + (px & 3);
        if (oy < a.H && ox < a.W)
            *reinterpret_cast<u32x4_t*>(obase + ((size_t)oy * a.W + ox) * a.out_stride + 8 * part) = *reinterpret_cast<const u32x4_t*>(stg + px * CO + 8 * part);
    }
    }
    }
    if constexpr (!BN_PERSIST) break;
    __syncthreads();                                        // every wave has left the staging / record areas of this tile
    if (tid == 0) s_next = next_tile;
    __syncthreads();
    lid = s_next;
    first_tile = false;
    if (lid < 0) break;
    }
    if (BN_PERSIST && tid == 0) {                           // the workgroup that leaves last re-arms the counter set for the next launch that uses it
        __threadfence();
        if (atomicAdd(a.ctr + 8, 1u) == gridDim.x - 1) {
            for (int i = 0; i < 9; ++i) atomicExch(a.ctr + i, 0u);
        }
    }
#ifdef MAF_BN_STAMPS
    BN_STAMP(6);                                            // epilogue
    if (a.prof && lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(a.prof + i, (unsigned long long)prof_acc[i]);
#endif
}

template <int K, int S1, int CT2, int C3T = 0, int NX = 0>
int launch_one(const BnArgs& a, hipStream_t s) {
    constexpr size_t lds = BnCfg<K, S1, CT2>::LDS;
    static_assert(lds <= 160 * 1024, "bottleneck: LDS budget");
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        int rc = maf_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_kernel<K, S1, CT2, C3T, NX>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(bottleneck)");
        if (rc) return rc;
        attr = true;
    }
    int grid = a.nwg;
    BnArgs b = a;
    if (BN_PERSIST) {
        static int per_cu = 0;
        static unsigned* ctrs = nullptr;                    // 64 counter sets handed out round-robin: launches in flight on different streams do not share one
        static unsigned seq = 0;
        if (!per_cu) {
            int n = 0;
            if (maf_check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&bottleneck_kernel<K, S1, CT2, C3T, NX>), 256, lds), "occupancy(bottleneck)")) return MAF_E_HIP;
            if (maf_check_hip(hipMalloc(reinterpret_cast<void**>(&ctrs), 64 * 16 * sizeof(unsigned)), "hipMalloc(bottleneck counters)")) return MAF_E_HIP;
            if (maf_check_hip(hipMemset(ctrs, 0, 64 * 16 * sizeof(unsigned)), "hipMemset(bottleneck counters)")) return MAF_E_HIP;
            per_cu = n > 0 ? n : 1;
        }
        const int slots = per_cu * 256;
        if (grid > slots) grid = slots;
        b.ctr = ctrs + (seq++ & 63) * 16;
    }
    hipLaunchKernelGGL((bottleneck_kernel<K, S1, CT2, C3T, NX>), dim3(grid), dim3(256), lds, s, b);
    return maf_check_hip(hipGetLastError(), "bottleneck launch");
}

template <int K>
int launch_k(const BnArgs& a, hipStream_t s) {
    const int s1 = (a.Cin + 31) / 32, ct2 = a.Cout <= 32 ? 2 : 4;
    if (a.w3) {
        // with the closing 1x1 of the RepHDW block: the shapes of MAF-YOLO-n (one bottleneck per block: NX = 2) and of the first blocks of s / m (NX = 3)
        const int c3t = a.C3 / 16;
#define BN_TAIL(KK, SS, CC, TT, NN) if constexpr (K == KK) { if (s1 == SS && ct2 == CC && c3t == TT && a.nx == NN) return launch_one<KK, SS, CC, TT, NN>(a, s); }
        BN_TAIL(3, 1, 2, 3, 2) BN_TAIL(5, 2, 4, 6, 2) BN_TAIL(5, 2, 4, 8, 2)
        BN_TAIL(3, 1, 2, 4, 3) BN_TAIL(5, 2, 4, 8, 3) BN_TAIL(3, 2, 4, 6, 3)
#undef BN_TAIL
        maf_set_error("bottleneck: no instantiation with the fused closing conv for this (k, c, C3, sources)");
        return MAF_E_UNSUPPORTED;
    }
    if (s1 == 1 && ct2 == 2) return launch_one<K, 1, 2>(a, s);          // c <= 32
    if (s1 == 2 && ct2 == 4) return launch_one<K, 2, 4>(a, s);          // 32 < c <= 64
    maf_set_error("bottleneck: unsupported channel count");
    return MAF_E_UNSUPPORTED;
}

}  // namespace

extern "C" int64_t maf_bottleneck_tail_record_bytes(int32_t c, int32_t nsrc, int32_t c3) {
    const int s1 = (c + 31) / 32, ys = c <= 32 ? 1 : 2, c3t = c3 / 16;
    if (c3 % 16 || nsrc < 2 || nsrc > 3) return 0;
    return (int64_t)(nsrc * s1 + ys) * c3t * 1024 + 64 * c3t;      /* A fragments [k-step][tile][64][8] f16 | b3 [4][c3t][4] f32 */
}

extern "C" int maf_bottleneck_tail_supported(int32_t k, int32_t c, int32_t nsrc, int32_t c3) {
    const int s1 = (c + 31) / 32, ct2 = c <= 32 ? 2 : 4, t = c3 / 16;
    if (c3 % 16 || c % 8 || c > 64) return 0;
    return (k == 3 && s1 == 1 && ct2 == 2 && t == 3 && nsrc == 2) || (k == 5 && s1 == 2 && ct2 == 4 && (t == 6 || t == 8) && nsrc == 2)
        || (k == 3 && s1 == 1 && ct2 == 2 && t == 4 && nsrc == 3) || (k == 5 && s1 == 2 && ct2 == 4 && t == 8 && nsrc == 3) || (k == 3 && s1 == 2 && ct2 == 4 && t == 6 && nsrc == 3);
}

extern "C" int64_t maf_bottleneck_record_bytes(int32_t k, int32_t cin, int32_t cout) {
    const int parts = k > 5 ? 2 : 1, s1 = (cin + 31) / 32, ct2 = cout <= 32 ? 2 : 4;
    return (int64_t)8 * k * parts * 16 * 16 + (int64_t)2 * s1 * 1024 + (int64_t)ct2 * 1024 + 256;   /* = BnCfg::REC */
}

int maf_launch_bottleneck(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->dtype == MAF_F16, "bottleneck: fp16 only (the fp32 parity mode runs the three kernels separately)");
    const maf_src_t& sr = op->src[0];
    const bool tail = op->nc > 0;
    MAF_REQUIRE((tail ? (op->nsrc == 2 || op->nsrc == 3) : op->nsrc == 1) && sr.mode == MAF_SRC_DIRECT && sr.ptr && op->out, "bottleneck: one direct source (nc > 0: + the other 1 or 2 concat sources of the closing conv)");
    MAF_REQUIRE(op->Cin % 8 == 0 && op->Cin <= 64 && op->Cout % 8 == 0 && op->Cout <= 64, "bottleneck: c <= 64 channels in and out, multiples of 8");
    MAF_REQUIRE(sr.stride % 8 == 0 && sr.coff % 8 == 0 && op->out_stride % 8 == 0 && op->out_coff % 8 == 0, "bottleneck: stride/offset alignment (16-byte pieces)");
    MAF_REQUIRE(op->act == MAF_ACT_SILU, "bottleneck: DepthBottleneckUni applies SiLU after every stage (common.py:918-927)");
    MAF_REQUIRE(op->tile_k > 0 && op->w && op->bias, "bottleneck: null weights (w = block records, bias = b2), tile_k = 32-channel mid blocks");
    BnArgs a;
    a.x = static_cast<const half_t*>(sr.ptr); a.out = static_cast<half_t*>(op->out);
    a.par = static_cast<const unsigned char*>(op->w); a.b2 = op->bias;
    a.B = op->B; a.H = op->H; a.W = op->W; a.Cin = op->Cin; a.Cout = op->Cout; a.nMB = op->tile_k;
    a.x_stride = sr.stride; a.x_coff = sr.coff; a.out_stride = op->out_stride; a.out_coff = op->out_coff;
    a.tilesX = maf_cdiv(a.W, 16); a.tilesY = maf_cdiv(a.H, 16);
    a.nwg = a.B * a.tilesX * a.tilesY;
    a.ntiles = a.nwg; a.ctr = nullptr;
    a.prof = nullptr; a.ko = 0;
    a.w3 = nullptr; a.C3 = 0; a.nx = 0;
    for (int i = 0; i < 3; ++i) { a.xs[i] = nullptr; a.xs_stride[i] = a.xs_coff[i] = 0; }
    if (tail) {
        // nc = C3: conv2(cat(src[1], .., src[nsrc - 1], src[0], y)) + SiLU of the surrounding RepHDW (common.py:944-946) applied before the store: out / out_stride /
        // out_coff describe ITS output (C3 channels), the bottleneck's own output is never written; aux[0] = record of maf_bottleneck_tail_record_bytes()
        MAF_REQUIRE(op->aux[0] && op->nc % 16 == 0, "bottleneck: nc > 0 needs aux[0] = the closing conv's record and nc a multiple of 16");
        a.nx = op->nsrc; a.C3 = op->nc; a.w3 = static_cast<const unsigned char*>(op->aux[0]);
        for (int i = 0; i < op->nsrc; ++i) {
            const maf_src_t& q = op->src[i == op->nsrc - 1 ? 0 : i + 1];          // concat order: the other sources first, this bottleneck's input last
            MAF_REQUIRE(q.ptr && q.mode == MAF_SRC_DIRECT && q.C == op->Cin && q.stride % 8 == 0 && q.coff % 8 == 0, "bottleneck: concat sources of the closing conv are direct, Cin channels each, 16-byte aligned");
            a.xs[i] = static_cast<const half_t*>(q.ptr); a.xs_stride[i] = q.stride; a.xs_coff[i] = q.coff;
        }
    }
#ifdef MAF_BN_PROFILE
    a.prof = const_cast<unsigned long long*>(static_cast<const unsigned long long*>(op->aux[3]));
    a.ko = (int)(intptr_t)op->aux[2];
#endif
    switch (op->ksize) {
        case 3: return launch_k<3>(a, s);
        case 5: return launch_k<5>(a, s);
        case 7: return launch_k<7>(a, s);
        case 9: return launch_k<9>(a, s);
        default: maf_set_error("bottleneck: k must be 3, 5, 7 or 9"); return MAF_E_UNSUPPORTED;
    }
}
