"""What a launch plan (engine.Plan) says about itself: the kernel (template instantiation) behind every op, its algorithmic bytes — inputs once + outputs once +
weights once, the byte model of SURVEY.md 8(d) applied to the FUSED graph — and its flops.  bench.py's roofline object and per-op table are built from these.
Split out of engine.py in round 6."""
from . import lib, pack

_ESIZE = {lib.F16: 2, lib.F32: 4}


def kernel_name(plan, idx):
    """Device kernel symbol (as rocprofv3 --kernel-trace prints it) behind op idx."""
    o = plan.ops[idx]
    T = "_Float16" if o.dtype == lib.F16 else "float"
    if o.kind in (lib.OP_CONV1X1, lib.OP_CONV3X3S2):
        if o.kind == lib.OP_CONV3X3S2:
            var = 3
        elif o.nsrc == 1 and o.src[0].mode == lib.SRC_POOL2:
            var = 2
        elif o.nsrc == 1 and o.src[0].mode == lib.SRC_DIRECT:
            var = 0
        else:
            var = 1
        outf32 = "true" if (o.out_f32 and o.dtype == lib.F16) else "false"
        if o.tile_k == 3:
            return "conv1x1_stream_kernel<%d, %d, %d>" % (o.tile_p, o.tile_c, -(-o.Cin // 32))
        if o.tile_k == 6:
            return "conv3s2_lds_kernel<%d, %d, 4, %d>" % (o.Cin, o.Cout, o.nc)
        if o.tile_k == 7:
            return "conv3s2_wreg_kernel<%d, %d, %d, %d, %d, %d>" % ((o.Cin, o.Cout) + pack.conv3x3_wreg_shape(o.Cin, o.Cout) + (2 if o.tile_p == 2 else 3, o.nc))
        if o.tile_k == 5:
            return "conv1x1_stream_lds_kernel<%d, %d, %s, %d, false>" % (o.tile_c, sum(-(-o.src[k].C // 32) for k in range(o.nsrc)), "true" if var == 1 else "false", 8 if o.tile_p == 2 else 4)      # (last: the statistics epilogue of the training form — never in a plan)
        return "conv_mfma_kernel<%s, %d, %d, %d, %s, %s, %s, %s>" % (T, o.tile_p, o.tile_c, var, outf32, "true" if o.tile_k == 4 else "false", "true" if o.tile_k in (2, 8) else "false",
                                                                     "true" if o.tile_k == 8 else "false")
    if o.kind == lib.OP_DWCONV:
        if o.tile_p == -1:
            return "dwconv_mfma_kernel<%d, %d>" % (o.ksize, o.act)
        if o.tile_p == -4:
            return "dwconv_p2_kernel<%d, %d, %d>" % (o.ksize, o.act, (o.Cout // o.Cin) if o.tile_k & 128 else 0)
        if o.tile_p == -2:
            return "dwconv_dot2_kernel<%d, 8, %d, %d>" % (o.ksize, 2 if (o.tile_k >> 8) % 2 == 0 else 1, o.act)
        return "dwconv_tile_kernel<%s, %d, %d>" % (T, o.ksize, o.act)
    if o.kind == lib.OP_BOTTLENECK:
        return "bottleneck_kernel<%d, %d, %d, %d, %d>" % (o.ksize, -(-o.Cin // 32), 2 if o.Cout <= 32 else 4, o.nc // 16, o.nsrc if o.nc else 0)
    if o.kind == lib.OP_CONV1DW:
        return "conv1dw_kernel<%d>" % o.ksize
    if o.kind == lib.OP_HEADTAIL:
        return "head_tail_kernel<%d, %d>" % (o.Cin, 2 if o.Cin <= 128 else 1)
    if o.kind == lib.OP_STEM2:
        return "stem2_kernel<%d, %d, %d>" % (o.ksize, o.Cout, o.nc)
    return {lib.OP_STEM: "stem_kernel", lib.OP_SPPF_POOL: "sppf_pool_lds_kernel", lib.OP_DECODE: "decode_kernel"}[o.kind]

def algorithmic_bytes(plan, idx):
    """Bytes one launch of op idx must move at minimum: every input element read once, every output
    element written once, weights once (SURVEY.md §8d layer-granular model)."""
    o = plan.ops[idx]
    es = plan.es
    px = plan.B * o.H * o.W
    if o.kind == lib.OP_STEM:
        ies = {lib.F16: 2, lib.F32: 4, lib.U8: 1}[o.in_dtype]
        return plan.B * 3 * o.Hin * o.Win * ies + px * o.Cout * es + 27 * o.Cout * 4
    if o.kind == lib.OP_STEM2:
        ies = {lib.F16: 2, lib.F32: 4, lib.U8: 1}[o.in_dtype]
        return plan.B * 3 * o.Hin * o.Win * ies + px * (o.nc or o.Cout) * es + (27 * o.ksize + 9 * o.ksize * o.Cout + o.Cout * o.nc) * es
    if o.kind == lib.OP_CONV1X1:
        rd = 0
        for i in range(o.nsrc):
            f = {lib.SRC_DIRECT: 1.0, lib.SRC_UP2: 0.25, lib.SRC_POOL2: 4.0}[o.src[i].mode]
            rd += px * f * o.src[i].C * es
        oes = 4 if o.out_f32 else es
        return int(rd) + px * o.Cout * oes + o.Cin * o.Cout * es
    if o.kind == lib.OP_CONV3X3S2:
        pooled = px * o.nc * es + o.Cin * o.nc * es if "pool1" in plan._ops[idx] else 0      # the pooled 1x1 branch of a one-launch MPRep: its output and weights (the input is the conv's)
        return (2 if "twin" in plan._ops[idx] else 1) * (plan.B * o.Hin * o.Win * o.Cin * es + px * o.Cout * es + 9 * o.Cin * o.Cout * es) + pooled
    if o.kind == lib.OP_DWCONV:
        return px * (o.Cin + o.Cout) * es + o.ksize * o.ksize * o.Cout * es
    if o.kind == lib.OP_CONV1DW:
        return px * (o.Cin + o.Cout) * es + (o.Cin * o.Cout + o.ksize * o.ksize * o.Cout) * es
    if o.kind == lib.OP_BOTTLENECK:
        mid = o.tile_k * 32
        wts = (o.Cin * mid + o.ksize * o.ksize * mid + mid * o.Cout) * es
        if o.nc:                                            # with the block's closing conv: every concat slot read once, its output written once
            return px * (o.nsrc * o.Cin + o.nc) * es + wts + (o.nsrc + 1) * o.Cin * o.nc * es
        return px * (o.Cin + o.Cout) * es + wts
    if o.kind == lib.OP_SPPF_POOL:
        return 4 * px * o.src[0].C * es
    if o.kind == lib.OP_HEADTAIL:                          # both branch inputs once, the prediction rows once, the two weight records
        return px * (2 * o.Cin * es + (5 + plan.nc) * 4) + 2 * (o.Cin * o.Cin + 80 * o.Cin) * es
    if o.kind == lib.OP_DECODE:
        anchors = sum(t.H * t.W for t, c, _ in plan.head_bufs if c is not None)          # levels a fused tail did not take
        return plan.B * anchors * ((plan.nc + 4 * (plan.reg_max + 1)) * 4 + (5 + plan.nc) * 4)
    return 0

def flops(plan, idx):
    o = plan.ops[idx]
    px = plan.B * o.H * o.W
    if o.kind == lib.OP_STEM:
        return 2 * px * 27 * o.Cout
    if o.kind == lib.OP_STEM2:
        return 2 * (4 * px * 27 * o.ksize + px * 9 * o.ksize * o.Cout + px * o.Cout * o.nc)
    if o.kind == lib.OP_CONV1X1:
        return 2 * px * o.Cin * o.Cout
    if o.kind == lib.OP_CONV3X3S2:
        return (2 if "twin" in plan._ops[idx] else 1) * 2 * px * 9 * o.Cin * o.Cout + (2 * px * o.Cin * o.nc if "pool1" in plan._ops[idx] else 0)
    if o.kind == lib.OP_DWCONV:
        return 2 * px * o.ksize * o.ksize * o.Cout
    if o.kind == lib.OP_CONV1DW:
        return 2 * px * (o.Cin * o.Cout + o.ksize * o.ksize * o.Cout)
    if o.kind == lib.OP_BOTTLENECK:
        mid = plan._ops[idx]["mid"]
        return 2 * px * (o.Cin * mid + o.ksize * o.ksize * mid + mid * o.Cout + (o.nsrc + 1) * o.Cin * o.nc)
    if o.kind == lib.OP_HEADTAIL:
        return 2 * px * (2 * o.Cin * o.Cin + o.Cin * (plan.nc + 4 * (plan.reg_max + 1)))
    return 0
