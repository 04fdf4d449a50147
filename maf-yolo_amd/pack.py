"""Host-side weight packing for the HIP kernels (done once per plan, fp32 algebra then one cast).

conv_mfma (csrc/conv_mfma.inc.h) takes the weights as the MFMA *B* operand in fragment order:

    packed[tile][step][lane = g*16 + p][j]      tile = n_tile*CT + ct,  g = lane >> 4, p = lane & 15
      f16: j < 8,  k = step*32 + g*8 + j        (v_mfma_f32_16x16x32_f16: lane holds B[8 consecutive k][col p])
      f32: j < 4,  k = step*16 + g*4 + j        (4 x v_mfma_f32_16x16x4_f32, MFMA j takes component j)
    output channel of MFMA column p of channel tile ct:   n_tile*16*CT + p*CT + ct

The column interleave makes the accumulator lane (g, p) own CT *consecutive* channels of 4 pixels and
the 16 lanes of a pixel cover 16*CT consecutive channels, so every store instruction of the epilogue
writes whole contiguous NHWC rows without an LDS transpose.  K is the concatenation of the sources'
channels (torch.cat order), each source zero-padded to whole k-steps; for 3x3 convs the steps run
tap-major (ky, kx), each tap zero-padded to whole k-steps.
"""
import torch

from . import lib


def tile_for(cout, m_pixels):
    """(tile_p, tile_c): least channel padding, then fewest channel tiles; 2 pixel tiles/wave when the grid stays big."""
    best = None
    for ct in (8, 6, 4, 2):
        n_tiles = -(-cout // (16 * ct))
        padded = n_tiles * 16 * ct
        key = (padded, n_tiles)
        if best is None or key < best[0]:
            best = (key, ct, n_tiles)
    ct, n_tiles = best[1], best[2]
    pt = 2 if (-(-m_pixels // 128)) * n_tiles >= 1024 else 1
    return pt, ct


def _steps(c, ks):
    return -(-c // ks)


def pack_matrix(w2d_segments, ct, dtype):
    """w2d_segments: list of [Cout, C_s] fp32 blocks (one per source / tap). Returns packed tensor (CPU)."""
    ks = 32 if dtype == lib.F16 else 16
    ch = ks // 4
    cout = w2d_segments[0].shape[0]
    n_tiles = -(-cout // (16 * ct))
    npad = n_tiles * 16 * ct
    cols = []
    for seg in w2d_segments:
        s = _steps(seg.shape[1], ks)
        blk = torch.zeros(npad, s * ks, dtype=torch.float32)
        blk[:cout, :seg.shape[1]] = seg
        cols.append(blk)
    wp = torch.cat(cols, 1)                                   # [Npad, S*KS]
    S = wp.shape[1] // ks
    # channel = n_tile*16CT + p*CT + ct ; k = step*KS + g*CH + j
    wp = wp.reshape(n_tiles, 16, ct, S, 4, ch)                # [nt, p, ct, S, g, j]
    wp = wp.permute(0, 2, 3, 4, 1, 5).contiguous()            # [nt, ct, S, g, p, j]
    wp = wp.reshape(n_tiles * ct, S, 64, ch)
    return wp.to(torch.float16 if dtype == lib.F16 else torch.float32).contiguous()


def pack_conv1x1(w, src_channels, ct, dtype):
    """w [Cout, Cin, 1, 1] fp32, Cin = sum(src_channels)."""
    w2 = w.reshape(w.shape[0], -1).float().cpu()
    segs, o = [], 0
    for c in src_channels:
        segs.append(w2[:, o:o + c])
        o += c
    assert o == w2.shape[1], "source channels do not add up to Cin"
    return pack_matrix(segs, ct, dtype)


def pack_conv3x3(w, ct, dtype):
    """w [Cout, Cin, 3, 3] fp32 -> tap-major segments."""
    w = w.float().cpu()
    return pack_matrix([w[:, :, ky, kx] for ky in range(3) for kx in range(3)], ct, dtype)


def pack_bias(b, ct):
    cout = b.shape[0]
    npad = -(-cout // (16 * ct)) * 16 * ct
    out = torch.zeros(npad, dtype=torch.float32)
    out[:cout] = b.float().cpu()
    return out


def pack_dw(w, dtype):
    """w [C, 1, k, k] -> [k*k, C] in the activation dtype."""
    c, _, k, _ = w.shape
    return w.float().cpu().reshape(c, k * k).t().contiguous().to(torch.float16 if dtype == lib.F16 else torch.float32)


def pack_dw_pairs(w):
    """w [C, 1, k, k] (C a multiple of 8) -> the weight pairs of csrc/dwconv_p2.hip, fp16 [C/8, k, 2 (half group h), 2 (set), (k+1)/2 (m), 4 (d), 2]:
    for channel 8 g + 4 h + d and kernel row ky, the EVEN set entry m = (w[ky][2m], w[ky][2m+1]), the ODD set entry m = (w[ky][2m-1], w[ky][2m]);
    taps outside 0..k-1 are zeros.  One dword per pair (low half = the tap of the even pixel of an input pixel pair)."""
    c, _, k, _ = w.shape
    assert c % 8 == 0
    npair = (k + 1) // 2
    wd = w.float().cpu().reshape(c // 8, 2, 4, k, k)                       # [g, h, d, ky, kx]
    wz = torch.zeros(c // 8, 2, 4, k, k + 3)
    wz[..., 1:k + 1] = wd                                                  # wz[..., u + 1] = w[..., u], zeros at u = -1, k, k + 1
    out = torch.zeros(c // 8, k, 2, 2, npair, 4, 2)
    for m in range(npair):
        for st, u0 in ((0, 2 * m), (1, 2 * m - 1)):
            out[:, :, :, st, m, :, 0] = wz[..., u0 + 1].permute(0, 3, 1, 2)          # [g, ky, h, d]
            out[:, :, :, st, m, :, 1] = wz[..., u0 + 2].permute(0, 3, 1, 2)
    return out.to(torch.float16).contiguous()


def pairs_from_nhwc(x):
    """[B, H, W, C] (W even) -> the pixel-pair layout [B, H, W/2, C, 2] (MAF_SRC_PAIRS): what a CONV1X1 with out_pairs = 1 writes."""
    b, h, w_, c = x.shape
    assert w_ % 2 == 0
    return x.reshape(b, h, w_ // 2, 2, c).transpose(3, 4).contiguous()


def pack_conv1dw(w1, b1, wdw, bdw):
    """Operands of MAF_OP_CONV1DW (csrc/conv1dw.hip), fp16: one record per 32 mid channels, in the order the kernel copies it to
    LDS: W1 fragments [2, S1, 64, 8] | b1 [32] f32 | Toeplitz table [8, k, parts, 16, 8] | bdw [32] f32.
    Inside a block the kernel's plane pl = 4s + g holds real channel 8g + s (lane group g ends with 8 consecutive channels), so the
    W1 columns / b1 of plane pl are those of channel 8*(pl % 4) + pl // 4; Toeplitz entry (s, G) and bdw are in real channel order."""
    mid, c = w1.shape[0], w1.shape[1]
    k = wdw.shape[-1]
    nmb = -(-mid // 32)
    mp = nmb * 32
    perm = torch.tensor([8 * (pl % 4) + pl // 4 for pl in range(32)])
    w1p = torch.zeros(mp, c); w1p[:mid] = w1.reshape(mid, c).float().cpu()
    b1p = torch.zeros(mp); b1p[:mid] = b1.float().cpu()
    w1p = w1p.reshape(nmb, 32, c)[:, perm].reshape(mp, c)
    b1q = b1p.reshape(nmb, 32)[:, perm]
    W1 = torch.cat([pack_matrix([w1p[m * 32:(m + 1) * 32]], 1, lib.F16) for m in range(nmb)], 0)       # [nmb*2, S1, 64, 8]
    wfull = torch.zeros(mp, 1, k, k); wfull[:mid] = wdw.float().cpu()
    toe = pack_dw_toeplitz(wfull)                                          # [nmb, 8, k, parts, 16, 8], entry (s, G) = channel 8G + s
    bdp = torch.zeros(mp); bdp[:mid] = bdw.float().cpu()

    def raw(t):
        return t.contiguous().view(torch.uint8).reshape(nmb, -1)
    rec = torch.cat([raw(W1.reshape(nmb, -1)), raw(b1q.contiguous()), raw(toe.reshape(nmb, -1)), raw(bdp.reshape(nmb, 32))], 1).contiguous()
    return rec, nmb


def pack_dw_toeplitz(w):
    """w [C, 1, k, k] -> the Toeplitz table of the matrix-core depth-wise kernel (csrc/dwconv_mfma.hip), fp16
    [C/32 blocks, 8 sets, k, parts, 16, 8]: entry (block cb, set s, tap row ky, window part, row i = 4G + r, element j) =
    w[cb*32 + 8G + s][ky][j - r + 4*part]; window 0 carries taps kx <= 4, window 1 (k > 5) the taps kx >= 5."""
    c, _, k, _ = w.shape
    ncb = -(-c // 32)
    wd = torch.zeros(ncb * 32, k, k); wd[:c] = w.float().cpu().reshape(c, k, k)
    parts = 2 if k > 5 else 1
    toe = torch.zeros(ncb, 8, k, parts, 4, 4, 8)                           # [cb, s, ky, part, G, r, j]
    wv = wd.reshape(ncb, 4, 8, k, k)                                       # [cb, G, s, ky, kx]
    for part in range(parts):
        for r in range(4):
            for j in range(8):
                kx = j - r + 4 * part
                if kx < 0 or kx >= k or (part == 0 and kx > 4) or (part == 1 and kx < 5):
                    continue
                toe[:, :, :, part, :, r, j] = wv[:, :, :, :, kx].permute(0, 2, 3, 1)          # [cb, s, ky, G]
    return toe.reshape(ncb, 8, k, parts, 16, 8).to(torch.float16).contiguous()


def pack_stem(w):
    """w [Cout, 3, 3, 3] -> fp32 [27, Cout], row = (c*3 + ky)*3 + kx."""
    return w.float().cpu().reshape(w.shape[0], 27).t().contiguous()


def pack_bottleneck(w1, b1, wdw, bdw, w2, b2):
    """Operands of MAF_OP_BOTTLENECK (csrc/bottleneck.hip), fp16: the mid channels are cut into 32-channel blocks.

    w1 [mid, c, 1, 1], wdw [mid, 1, k, k], w2 [c, mid, 1, 1] (+ fp32 biases).  Returns (records, b2 pad, n_blocks, ct2):
    `records` is a uint8 tensor [n_blocks, REC]; one record = everything the kernel needs for one 32-channel mid block, in the
    order it copies it to LDS:  W1 fragments [2, S1, 64, 8] f16 | b1 [32] f32 | Toeplitz table [8, k, parts, 16, 8] f16 |
    W2 fragments [ct2, 64, 8] f16 | bdw [32] f32.

    Toeplitz entry (block mb, set s, tap row ky, window part, row i = 4G + r, element j) = wdw[mb*32 + 4s + G][ky][j - r + 4*part]
    — the depth-wise filter row of channel 4s + G (set s convolves the four channels 4s .. 4s+3; bdw and the rows of W2 are
    permuted to the kernel's k index 8G + s accordingly) seen from output column r of a 4-output group through an 8-wide input
    window; window 0 carries taps kx <= 4, window 1 (k > 5 only, shifted 4 columns) the taps kx >= 5."""
    mid, c = w1.shape[0], w1.shape[1]
    k = wdw.shape[-1]
    nmb = -(-mid // 32)
    mp = nmb * 32
    # The kernel evaluates SiLU as v * rcp(1 + exp2(-v)) on v = log2(e) * (pre-activation): the multiplication by log2(e) that exp() needs
    # is folded into the weights — W1, b1 and bdw carry the factor, T1 and T2 live in LDS / registers as log2(e) * their value (the
    # depth-wise conv is linear, so its filters stay), and W2 carries 1 / log2(e).  One vector instruction less per SiLU in a VALU-bound kernel.
    L2E = 1.4426950408889634
    w1, b1, bdw, w2 = w1.float() * L2E, b1.float() * L2E, bdw.float() * L2E, w2.float() / L2E
    w1p = torch.zeros(mp, c); w1p[:mid] = w1.reshape(mid, c).float().cpu()
    W1 = torch.cat([pack_matrix([w1p[m * 32:(m + 1) * 32]], 1, lib.F16) for m in range(nmb)], 0)       # [nmb*2, S1, 64, 8]
    b1p = torch.zeros(mp); b1p[:mid] = b1.float().cpu()
    wd = torch.zeros(mp, k, k); wd[:mid] = wdw.float().cpu().reshape(mid, k, k)
    parts = 2 if k > 5 else 1
    toe = torch.zeros(nmb, 8, k, parts, 4, 4, 8)                           # [mb, s, ky, part, G, r, j]
    wv = wd.reshape(nmb, 8, 4, k, k)                                       # [mb, s, G, ky, kx]
    for part in range(parts):
        for r in range(4):
            for j in range(8):
                kx = j - r + 4 * part
                if kx < 0 or kx >= k or (part == 0 and kx > 4) or (part == 1 and kx < 5):
                    continue
                toe[:, :, :, part, :, r, j] = wv[:, :, :, :, kx].permute(0, 1, 3, 2)          # [mb, s, ky, G]
    bdp = torch.zeros(mp); bdp[:mid] = bdw.float().cpu()
    bdp = bdp.reshape(nmb, 8, 4).permute(0, 2, 1).contiguous().reshape(mp)     # [mb][G][s]: lane group G reads its 8 biases contiguously
    ct2 = 2 if c <= 32 else 4
    w2f = torch.zeros(c, mp); w2f[:, :mid] = w2.reshape(c, mid).float().cpu()
    w2f = w2f.reshape(c, nmb, 8, 4).permute(0, 1, 3, 2).reshape(c, mp)         # k index 8G + s <- channel 4s + G
    W2 = torch.cat([pack_matrix([w2f[:, m * 32:(m + 1) * 32]], ct2, lib.F16) for m in range(nmb)], 0)   # [nmb*ct2, 1, 64, 8]
    b2p = torch.zeros(16 * ct2); b2p[:c] = b2.float().cpu()
    def raw(t):
        return t.contiguous().view(torch.uint8).reshape(nmb, -1)
    rec = torch.cat([raw(W1.reshape(nmb, -1)), raw(b1p.reshape(nmb, 32)), raw(toe.reshape(nmb, -1).to(torch.float16)),
                     raw(W2.reshape(nmb, -1)), raw(bdp.reshape(nmb, 32))], 1).contiguous()
    return rec, b2p, nmb, ct2


def pack_bottleneck_tail(w3, b3, c, nsrc):
    """Record of the closing 1x1 conv fused behind MAF_OP_BOTTLENECK (op.nc = C3 > 0, csrc/bottleneck.hip): RepHDW.conv2 (common.py:944-946) over
    cat(slot 0, .., slot nsrc-1, y) with c channels per slot, w3 [C3, (nsrc + 1) c, 1, 1], C3 a multiple of 16.

    Layout: A fragments [k-step][tile t3][64 lanes][8] f16 | b3 [4 g][C3 / 16][4 rr] f32.  The GEMM runs transposed (rows = output channels):
    row m = 4g + rr of tile t3 is output value v = rr * (C3 / 16) + t3 of lane group g, i.e. channel (v // P) * 4P + g * P + v % P with P = 8 (4 when a
    lane's 4 C3 / 16 values are no multiple of 8): the four lane groups of one store instruction cover a contiguous 4P-channel run of a pixel.
    k-steps: S1 = ceil(c / 32) per slot (k-slot (g, j) = channel 32 ks + 8g + j of the slot, zero past c), then the y part straight from the transposed
    accumulators of the bottleneck's second 1x1: c > 32 (CT2 = 4) two steps with k-slot (g, j) of step j2 = channel 16g + 8 j2 + j, else one step with 8g + j."""
    C3 = w3.shape[0]
    assert C3 % 16 == 0 and w3.reshape(C3, -1).shape[1] == (nsrc + 1) * c and c % 8 == 0 and c <= 64
    c3t, s1, ct2 = C3 // 16, -(-c // 32), (2 if c <= 32 else 4)
    ys = ct2 // 2
    nv = 4 * c3t
    P = 8 if nv % 8 == 0 else 4
    w = w3.detach().reshape(C3, -1).float().cpu()
    cols = []
    for s_ in range(nsrc):
        for ks in range(s1):
            for gl in range(4):
                for j in range(8):
                    ch = 32 * ks + 8 * gl + j
                    cols.append(s_ * c + ch if ch < c else -1)
    for j2 in range(ys):
        for gl in range(4):
            for j in range(8):
                ch = 16 * gl + 8 * j2 + j if ct2 == 4 else 8 * gl + j
                cols.append(nsrc * c + ch if ch < c else -1)
    cols = torch.tensor(cols)
    wk = torch.zeros(C3, cols.numel())
    ok = cols >= 0
    wk[:, ok] = w[:, cols[ok]]
    g_, rr_, t_ = torch.meshgrid(torch.arange(4), torch.arange(4), torch.arange(c3t), indexing="ij")
    v = rr_ * c3t + t_
    ch_out = (v // P) * (4 * P) + g_ * P + v % P                         # [g][rr][t3]
    assert sorted(ch_out.reshape(-1).tolist()) == list(range(C3))
    nst = cols.numel() // 32
    rows = ch_out.permute(2, 0, 1).reshape(c3t, 16)                      # [t3][m = 4g + rr]
    fr = wk[rows.reshape(-1)].reshape(c3t, 16, nst, 4, 8)                # [t3][m][step][gl][j]
    fr = fr.permute(2, 0, 3, 1, 4).contiguous().half()                  # [step][t3][gl][m][j]: lane = gl * 16 + m
    b3p = b3.detach().float().cpu()[ch_out.permute(0, 2, 1).reshape(-1)].contiguous()      # [g][t3][rr]
    rec = torch.cat([fr.reshape(-1).view(torch.uint8), b3p.view(torch.uint8)])
    assert rec.numel() == nst * c3t * 1024 + 64 * c3t
    return rec


def pack_head_tail(w1, b1, w2, b2):
    """Weight record of one branch of MAF_OP_HEADTAIL (csrc/head_tail.hip): 1x1 conv w1 [C,C] (+ b1, SiLU) followed by 1x1 conv w2
    [n2 <= 80, C] (+ b2).  Layout: A fragments of W1 [C/16][C/32][64 lanes][8] f16 (lane (g, i): output channel 16t + i, input
    channels 32ks + 8g ..+7) | B fragments of W2 [5][C/32][64][8] f16 whose K axis follows the accumulator layout of the first GEMM
    (lane (g, n): output column 16 t2 + n; k-slot q of step j = channel 32j + 4g + q for q < 4, 32j + 16 + 4g + q - 4 otherwise)
    | b1 fp32 [C] | b2 fp32 [80] (zero padded)."""
    w1 = w1.detach().float().cpu().reshape(w1.shape[0], -1)
    w2 = w2.detach().float().cpu().reshape(w2.shape[0], -1)
    C = w1.shape[0]
    assert w1.shape == (C, C) and w2.shape[1] == C and w2.shape[0] <= 80 and C % 32 == 0
    ks, t1 = C // 32, C // 16
    f1 = w1.view(t1, 16, ks, 4, 8).permute(0, 2, 3, 1, 4).contiguous().half()                  # [t][ks][g][i][j]
    w2p = torch.zeros(80, C)
    w2p[:w2.shape[0]] = w2
    j, g, q = torch.meshgrid(torch.arange(ks), torch.arange(4), torch.arange(8), indexing="ij")
    ch = 32 * j + torch.where(q < 4, 4 * g + q, 16 + 4 * g + q - 4)                            # [ks][4][8]
    f2 = w2p[:, ch.reshape(-1)].view(5, 16, ks, 4, 8).permute(0, 2, 3, 1, 4).contiguous().half()   # [t2][j][g][n][q]
    b2p = torch.zeros(80)
    b2p[:b2.numel()] = b2.detach().float().cpu()
    rec = torch.cat([f1.reshape(-1).view(torch.uint8), f2.reshape(-1).view(torch.uint8),
                     b1.detach().float().cpu().contiguous().view(torch.uint8), b2p.view(torch.uint8)])
    assert rec.numel() == C * C * 2 + 80 * C * 2 + C * 4 + 320
    return rec


def pack_stem2(w0, b0, w1, b1, w3=None, b3=None):
    """Record of MAF_OP_STEM2 (csrc/stem2.hip): backbone.0 w0 [C0,3,3,3] + b0 and backbone.1 w1 [C1,C0,3,3] + b1, both in deploy form; optionally
    the 1x1 conv w3 [C3,C1] + b3 that follows (appended: fragments [C3/16][ceil(C1/32)][64][8] f16 — lane (g, i): output channel 16t + i,
    k-slot q of step j = channel 32j + 4g + q for q < 4, 32j + 16 + 4g + q - 4 otherwise, the order the accumulators of conv 1 come in —
    then b3 fp32 [C3]).
    Layout: B fragments of conv 0 [NT0 = 2 (C0 <= 32) or 3 (C0 = 48) tiles][64 lanes][8] f16 (lane (g, n): tap k = 8g + j = (c*3 + ky)*3 + kx, output channel 16t + n;
    zero for k >= 27 and channels >= C0) | B fragments of conv 1 [ceil(9*C0/8 / 4)][C1/16][64][8] f16 (lane (g, n) of k-step s: pair
    q = 4s + g -> tap q // (C0/8), channel group q % (C0/8); element j = input channel 8*group + j; output channel 16t + n; zero for
    pairs past the last tap) | b0 fp32 [16 NT0] (zero padded) | b1 fp32 [C1]."""
    w0 = w0.detach().float().cpu(); w1 = w1.detach().float().cpu()
    C0, C1 = w0.shape[0], w1.shape[0]
    assert w0.shape[1:] == (3, 3, 3) and w1.shape[1:] == (C0, 3, 3) and C0 % 8 == 0 and C0 <= 48 and C1 % 16 == 0
    nt0 = 2 if C0 <= 32 else 3                                        # 16-channel tiles of conv 0 (n: 24, s: 32 -> 2; m: 48 -> 3, round 6)
    m0 = torch.zeros(32, 16 * nt0)                                    # [k][channel]
    m0[:27, :C0] = w0.reshape(C0, 27).t()
    f0 = m0.view(4, 8, nt0, 16).permute(2, 0, 3, 1).contiguous().half()         # [t][g][n][j]
    gr = C0 // 8
    npair = 9 * gr
    ks1 = (npair + 3) // 4
    m1 = torch.zeros(ks1 * 4, 8, C1)                                 # [pair][j][out channel]
    for q in range(npair):
        tap, grp = divmod(q, gr)
        m1[q] = w1[:, 8 * grp:8 * grp + 8, tap // 3, tap % 3].t()
    f1 = m1.view(ks1, 4, 8, C1 // 16, 16).permute(0, 3, 1, 4, 2).contiguous().half()   # [s][t][g][n][j]
    b0p = torch.zeros(16 * nt0); b0p[:C0] = b0.detach().float().cpu()
    parts = [f0.reshape(-1).view(torch.uint8), f1.reshape(-1).view(torch.uint8), b0p.view(torch.uint8), b1.detach().float().cpu().contiguous().view(torch.uint8)]
    n = nt0 * 1024 + ks1 * (C1 // 16) * 1024 + nt0 * 64 + C1 * 4
    if w3 is not None:
        w3 = w3.detach().float().cpu().reshape(w3.shape[0], -1)
        C3 = w3.shape[0]
        assert w3.shape[1] == C1 and C3 % 16 == 0
        ks3 = (C1 + 31) // 32
        w3p = torch.zeros(C3, ks3 * 32)
        w3p[:, :C1] = w3
        j, g, q = torch.meshgrid(torch.arange(ks3), torch.arange(4), torch.arange(8), indexing="ij")
        ch = 32 * j + torch.where(q < 4, 4 * g + q, 16 + 4 * g + q - 4)
        f3 = w3p[:, ch.reshape(-1)].view(C3 // 16, 16, ks3, 4, 8).permute(0, 2, 3, 1, 4).contiguous().half()   # [t][j][g][i][q]
        parts += [f3.reshape(-1).view(torch.uint8), b3.detach().float().cpu().contiguous().view(torch.uint8)]
        n += ks3 * (C3 // 16) * 1024 + C3 * 4
    rec = torch.cat(parts)
    assert rec.numel() == n
    return rec


def pack_conv3x3_lds(w, b):
    """Record of the LDS-resident 3x3 stride-2 conv (csrc/conv3s2_lds.hip, tile_k = 6): w [Cout, Cin, 3, 3], b [Cout] -> fragments
    [ceil(9*Cin/8 / 4)][Cout/16][64 lanes][8] f16 (lane (g, n) of k-step s: pair q = 4s + g -> tap q // (Cin/8), channel group q % (Cin/8);
    element j = input channel 8*group + j; output channel 16t + n; zero for pairs past the last tap) | bias fp32 [Cout]."""
    w = w.detach().float().cpu()
    cout, cin = w.shape[:2]
    assert cin % 8 == 0 and cout % 16 == 0 and w.shape[2:] == (3, 3)
    gr = cin // 8
    npair = 9 * gr
    ks = (npair + 3) // 4
    m1 = torch.zeros(ks * 4, 8, cout)                               # [pair][j][out channel]
    for q in range(npair):
        tap, grp = divmod(q, gr)
        m1[q] = w[:, 8 * grp:8 * grp + 8, tap // 3, tap % 3].t()
    f = m1.view(ks, 4, 8, cout // 16, 16).permute(0, 3, 1, 4, 2).contiguous().half()   # [s][t][g][n][j]
    rec = torch.cat([f.reshape(-1).view(torch.uint8), b.detach().float().cpu().contiguous().view(torch.uint8)])
    assert rec.numel() == ks * (cout // 16) * 1024 + cout * 4
    return rec


def pack_mprep_lds(w, b, w1, b1):
    """Record of MPRep in one launch (csrc/conv3s2_lds.hip, tile_k = 6 with nc = C1: cat(conv1(MaxPool2d(2, 2)(x)), conv2(x)), common.py:776-792):
    pack_conv3x3_lds(w, b) of conv2, then conv1 — w1 [C1, Cin, 1, 1], b1 [C1] — as fragments [ceil(Cin / 32)][C1 / 16][64 lanes][8] f16 (lane (g, n)
    of k-step s: output channel 16t + n, input channels 32s + 8g .. + 7, zero past Cin) | bias fp32 [C1]."""
    w1 = w1.detach().float().cpu().reshape(w1.shape[0], -1)
    c1, cin = w1.shape
    assert c1 % 16 == 0 and cin == w.shape[1]
    ks1 = -(-cin // 32)
    m1 = torch.zeros(ks1 * 32, c1)
    m1[:cin] = w1.t()
    f = m1.view(ks1, 4, 8, c1 // 16, 16).permute(0, 3, 1, 4, 2).contiguous().half()   # [s][t][g][n][j]
    rec = torch.cat([pack_conv3x3_lds(w, b), f.reshape(-1).view(torch.uint8), b1.detach().float().cpu().contiguous().view(torch.uint8)])
    return rec


_WREG_SHAPES = {(128, 128): (8, 1), (96, 96): (6, 1), (96, 64): (4, 2), (64, 64): (4, 2), (64, 96): (6, 1), (128, 96): (6, 1)}        # (Cin, Cout) -> (waves along the channels, waves along the pixels)


def conv3x3_wreg_shape(cin, cout):
    """(waves along the channels, waves along the pixels) of the register-resident 3x3 stride-2 conv (csrc/conv3s2_wreg.hip, tile_k = 7), or None."""
    return _WREG_SHAPES.get((cin, cout))


def pack_mprep_wreg(w, b, w1, b1):
    """Record of MPRep in one launch on the register-resident 3x3 kernel (csrc/conv3s2_wreg.hip, tile_k = 7 with nc = C1; 96 -> 96 + 96): pack_conv3x3_wreg(w, b)
    of conv2, then conv1 — w1 [C1, Cin, 1, 1], b1 [C1] — as fragments [C1 / 16 channel tiles][Cin / 32 k-steps][64 lanes][8] f16 (lane (g, n) of k-step j of
    tile t: output channel 16t + n, input channels 32j + 8g .. + 7) | bias fp32 [C1]."""
    w1 = w1.detach().float().cpu().reshape(w1.shape[0], -1)
    c1, cin = w1.shape
    assert c1 % 16 == 0 and cin % 32 == 0 and cin == w.shape[1]
    f = w1.t().contiguous().view(cin // 32, 4, 8, c1 // 16, 16).permute(3, 0, 1, 4, 2).contiguous().half()   # [j][g][e][t][n] -> [t][j][g][n][e]
    return torch.cat([pack_conv3x3_wreg(w, b), f.reshape(-1).view(torch.uint8), b1.detach().float().cpu().contiguous().view(torch.uint8)])


def pack_conv3x3_wreg(w, b):
    """Record of the register-resident-weight 3x3 stride-2 conv (csrc/conv3s2_wreg.hip, tile_k = 7): w [Cout, Cin, 3, 3], b [Cout] -> fragments
    [Cout / 16 channel tiles][9 * Cin / 32 k-steps][64 lanes][8] f16 — lane (g, n) of k-step s of tile t: output channel 16 t + n, pair q = 4 s + g ->
    tap q // (Cin / 8) (tap-major: uniform per k-step since Cin % 32 == 0), channel group q % (Cin / 8), element j = input channel 8 * group + j —
    followed by the bias, fp32 [Cout]."""
    w = w.detach().float().cpu()
    cout, cin = w.shape[:2]
    assert (cin, cout) in _WREG_SHAPES and cin % 32 == 0 and cout % 16 == 0 and w.shape[2:] == (3, 3)
    gr = cin // 8
    ks = 9 * cin // 32
    m1 = torch.zeros(ks * 4, 8, cout)                               # [pair][j][out channel]
    for q in range(9 * gr):
        tap, grp = divmod(q, gr)
        m1[q] = w[:, 8 * grp:8 * grp + 8, tap // 3, tap % 3].t()
    # [s][g][j][t][n] -> [t][s][g][n][j]
    f = m1.view(ks, 4, 8, cout // 16, 16).permute(3, 0, 1, 4, 2).contiguous().half()
    rec = torch.cat([f.reshape(-1).view(torch.uint8), b.detach().float().cpu().contiguous().view(torch.uint8)])
    assert rec.numel() == (cout // 16) * ks * 1024 + cout * 4
    return rec
