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


def pack_stem(w):
    """w [Cout, 3, 3, 3] -> fp32 [27, Cout], row = (c*3 + ky)*3 + kx."""
    return w.float().cpu().reshape(w.shape[0], 27).t().contiguous()


def pack_bottleneck(w1, b1, wdw, bdw, w2, b2):
    """Operands of MAF_OP_BOTTLENECK (csrc/bottleneck.hip), fp16: the mid channels are cut into 64-channel blocks.

    w1 [mid, c, 1, 1], wdw [mid, 1, k, k], w2 [c, mid, 1, 1] (+ fp32 biases).  Returns (W1 packed, b1 pad, wdw [k*k][mid_pad],
    bdw pad, W2 packed per block, b2 pad, n_blocks, ct2)."""
    mid, c = w1.shape[0], w1.shape[1]
    k = wdw.shape[-1]
    nmb = -(-mid // 64)
    mp = nmb * 64
    w1p = torch.zeros(mp, c); w1p[:mid] = w1.reshape(mid, c).float().cpu()
    W1 = pack_matrix([w1p], 4, lib.F16)                                   # [nmb*4, steps1, 64, 8]
    b1p = torch.zeros(mp); b1p[:mid] = b1.float().cpu()
    wd = torch.zeros(k * k, mp); wd[:, :mid] = wdw.float().cpu().reshape(mid, k * k).t()
    bdp = torch.zeros(mp); bdp[:mid] = bdw.float().cpu()
    ct2 = 2 if c <= 32 else 4
    w2f = torch.zeros(c, mp); w2f[:, :mid] = w2.reshape(c, mid).float().cpu()
    W2 = torch.cat([pack_matrix([w2f[:, m * 64:(m + 1) * 64]], ct2, lib.F16) for m in range(nmb)], 0)   # [nmb*ct2, 2, 64, 8]
    b2p = torch.zeros(16 * ct2); b2p[:c] = b2.float().cpu()
    return W1.contiguous(), b1p, wd.to(torch.float16).contiguous(), bdp, W2.contiguous(), b2p, nmb, ct2
