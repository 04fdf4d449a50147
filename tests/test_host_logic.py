"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the graph /
re-param / packing logic is right, plans build (no launches without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import maf_yolo_amd as M
from maf_yolo_amd import lib, pack, arch
from maf_yolo_amd.engine import Plan
from oracle import maf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return lib.load()


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "mafyolo_hip.h")).read()
    declared = set(re.findall(r"\b(maf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS)
    for s in declared:
        assert hasattr(built, s), s
    assert built.maf_version() >= 100


def test_struct_layout_matches_header(built):
    # sizeof(maf_op_t) as laid out by the C compiler == ctypes mirror (guards against field drift)
    src = '#include <stdio.h>\n#include "mafyolo_hip.h"\nint main(){printf("%zu %zu", sizeof(maf_src_t), sizeof(maf_op_t));return 0;}'
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        a, b = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert int(a) == ctypes.sizeof(lib.MafSrc) and int(b) == ctypes.sizeof(lib.MafOp)


def test_error_reporting_without_gpu(built):
    op = lib.MafOp()
    op.kind = 99
    rc = built.maf_op_launch(ctypes.byref(op), None)
    assert rc == -2 and b"unknown op kind" in built.maf_last_error()
    op.kind = lib.OP_CONV1X1
    op.dtype = 7
    assert built.maf_op_launch(ctypes.byref(op), None) == -1
    with pytest.raises(lib.MafError):
        lib.check(built.maf_nms(None, 1, 1, 1, 0.5, 0.5, None, 0, 0, 0, 300, None, 0, None, None, None, None))


@pytest.mark.parametrize("scale", ["n", "s", "m"])
def test_model_state_dict_and_fused_weights(scale):
    m = M.Model(scale)
    sd = O.synth_state_dict(scale, 0)
    assert [k for k, _ in O.state_spec(scale)] == list(m.state_dict().keys())
    m.load_state_dict(sd, strict=True)
    m.eval()
    dw = O.reparam(sd, scale)
    with torch.no_grad():
        n = 0
        for name, (w, b) in dw.items():
            obj = m
            for part in name.split("."):
                obj = obj[int(part)] if part.isdigit() else getattr(obj, part, None)
                if obj is None:
                    break
            parent = m
            parts = name.split(".")
            # fused() lives on the owning block: strip the deploy-form suffix
            for suffix, up in ((".rbr_reparam", 1), (".dwconv.lk_origin", 2), (".conv", 1)):
                if name.endswith(suffix):
                    parts = name[: -len(suffix)].split(".")
                    break
            else:
                continue            # plain pred convs carry no BN
            for part in parts:
                parent = parent[int(part)] if part.isdigit() else getattr(parent, part)
            fw, fb = parent.fused()
            assert torch.allclose(fw, w, atol=1e-6) and torch.allclose(fb, b, atol=1e-6), name
            n += 1
        assert n == len(dw) - 6
        x = O.synth_images(1, 64, 1)
        heads = m._forward_train_form(x)
        for h, oh in zip(heads, O.forward_train_form(sd, scale, x)):
            for a, b in zip(h, oh):
                assert torch.equal(a, b)


def test_reference_yaml_schema_roundtrip():
    for s in "nsm":
        nodes = arch.builtin(s)
        assert [n.cout for n in nodes][:34] == O.channels_out(s)
    with pytest.raises(NotImplementedError):
        arch.nodes_from_yaml_dict(dict(depth_multiple=1, width_multiple=1, backbone=[[-1, 1, "Focus", [64]]], effidehead=[]))


def test_train_mode_output_contract():
    m = M.Model("n").train()
    out, feats = m(torch.rand(2, 3, 64, 64))
    f, cls, reg = out
    assert cls.shape == (2, 8 * 8 + 4 * 4 + 2 * 2, 80) and reg.shape == (2, 84, 68) and len(f) == 3 and len(feats) == 3
    loss = cls.sum() + reg.sum()
    loss.backward()
    assert m.backbone[0].rbr_dense.conv.weight.grad is not None


def test_cpu_input_fails_loudly():
    m = M.Model("n").eval()
    with pytest.raises(M.MafError):
        m(torch.rand(1, 3, 64, 64))
    with pytest.raises(M.MafError):
        M.non_max_suppression(torch.rand(1, 10, 85))
    with pytest.raises(AssertionError):
        M.non_max_suppression(torch.rand(1, 10, 85), conf_thres=1.5)


def test_pack_layout():
    # packed[tile][step][lane][j] == W[chan(tile,lane)][k(step,lane,j)]
    cout, cin, ct = 72, 48, 4
    w = torch.arange(cout * cin, dtype=torch.float32).reshape(cout, cin, 1, 1)
    for dt, ks, ch in ((lib.F16, 32, 8), (lib.F32, 16, 4)):
        p = pack.pack_conv1x1(w, [24, 24], ct, dt).float()
        steps = 2 * -(-24 // ks)
        assert p.shape == (2 * ct, steps, 64, ch)
        for tile, step, lane, j in ((0, 0, 0, 0), (1, 1, 17, 3), (4, steps - 1, 63, ch - 1), (7, 0, 37, 1)):
            nt, c_t = divmod(tile, ct)
            g, i = lane >> 4, lane & 15
            chan = nt * 16 * ct + i * ct + c_t
            spp = -(-24 // ks)                       # steps per source
            src, ls = divmod(step, spp)
            kk = ls * ks + g * ch + j
            exp = w[chan, src * 24 + kk, 0, 0].item() if (chan < cout and kk < 24) else 0.0
            assert p[tile, step, lane, j].item() == (float(np.float16(exp)) if dt == lib.F16 else exp)
    assert pack.tile_for(24, 10 ** 6)[1] == 2 and pack.tile_for(128, 10 ** 6) == (2, 8) and pack.tile_for(384, 12800)[0] == 1


@pytest.mark.parametrize("scale,nops32", [("n", 88 + 2), ("s", 118 + 2), ("m", 148 + 2)])
def test_plan_builds_on_cpu(built, scale, nops32):
    nops = nops32 - 3              # fp16 plans: cls_conv + reg_conv of a level are ONE two-filter depth-wise launch (fp32 parity plans keep them apart)
    m = M.Model(scale).eval()
    m.fuse_head = False
    m.fuse_stem = False
    m.fuse_mprep = False           # (checked at the end)
    tw = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"), fuse=False)       # default: the equal side convs 23 / 24 and 27 / 28 as twin launches
    pairs = ["backbone.23.block+backbone.24.block", "backbone.27.block+backbone.28.block"][0 if scale != "m" else 1:]    # m: nodes 22 and 20 differ in width
    assert len(tw.ops) == nops - len(pairs) and [n for n in tw.op_names if "+" in n] == pairs
    for o, name in zip(tw.ops, tw.op_names):
        assert bool(o.aux[0]) == ("+" in name) or o.kind == lib.OP_DWCONV, name
        if "+" in name:
            assert all(tw._abase <= o.aux[k] < tw._abase + tw._arena_size for k in (0, 3)) and o.aux[3] != o.out and o.aux[0] != o.src[0].ptr
    m.twin_convs = False
    plan = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"), fuse=False)
    assert len(plan.ops) == nops and built.maf_engine_num_ops(plan._engine) == nops
    lo, hi = plan._abase, plan._abase + plan._arena_size
    for o, name in zip(plan.ops, plan.op_names):
        if o.kind in (lib.OP_CONV1X1, lib.OP_CONV3X3S2, lib.OP_DWCONV, lib.OP_BOTTLENECK, lib.OP_CONV1DW):
            assert sum(o.src[i].C for i in range(o.nsrc)) == o.Cin, name
            for i in range(o.nsrc):
                assert lo <= o.src[i].ptr < hi and o.src[i].stride % 8 == 0 and o.src[i].coff % 8 == 0, name
            assert lo <= o.out < hi, name
    assert plan.A == 8 * 8 + 4 * 4 + 2 * 2
    m.fuse_bottlenecks = True                      # fused DepthBottleneckUni: 3 launches -> 1 wherever c <= 64, -> 2 (conv1+dw | 1x1) elsewhere
    tails = {"n": ["backbone.2.m.0+conv2", "backbone.4.m.0+conv2", "backbone.20.m.0+conv2", "backbone.22.m.0+conv2"],
             "s": ["backbone.2.m.1+conv2", "backbone.4.m.1+conv2"], "m": ["backbone.2.m.1+conv2"]}[scale]
    one = tails if scale == "n" else []
    assert [n for n in Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu")).op_names if n.endswith("+conv2")] == one   # "auto" (default): blocks of ONE bottleneck only — every block of n
    from maf_yolo_amd.config import cfg, Config
    for val, want in ((3, tails), (1, one), (0, [])):                   # cfg.fuse_tail (MAF_FUSE_TAIL): 3 = "auto" takes every instantiation (s / m opt-in), 1 = the default rule, 0 = off
        keep, cfg.fuse_tail = cfg.fuse_tail, val
        try:
            assert [n for n in Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu")).op_names if n.endswith("+conv2")] == want, val
        finally:
            cfg.fuse_tail = keep
    assert Config({"MAF_FUSE_TAIL": "3", "MAF_STEP_TAPE": "0"}).overridden() == {"fuse_tail": 3, "step_tape": False} and Config({}).overridden() == {}
    m.fuse_tail = True
    wt = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"))                  # the block's closing conv inside its last bottleneck's launch
    assert [n for n in wt.op_names if n.endswith("+conv2")] == tails and not any(n[:-len(".m.0+conv2")] + ".conv2" in wt.op_names for n in tails)
    for o, name in zip(wt.ops, wt.op_names):
        if name in tails:
            depth = int(name.split(".m.")[1][0]) + 1
            assert o.kind == lib.OP_BOTTLENECK and o.nc % 16 == 0 and o.nsrc == depth + 1 and o.aux[0] and all(o.src[i].C == o.Cin for i in range(o.nsrc))
            assert built.maf_bottleneck_tail_supported(o.ksize, o.Cin, o.nsrc, o.nc) == 1 and len({(o.src[i].ptr, o.src[i].coff) for i in range(o.nsrc)}) == o.nsrc
    m.fuse_tail = False
    fused = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"))
    assert len(wt.ops) == len(fused.ops) - len(tails)
    full, part = {"n": (6, 4), "s": (4, 16), "m": (2, 28)}[scale]
    assert len(fused.ops) == nops - 2 * full - part
    assert sum(1 for o in fused.ops if o.kind == lib.OP_BOTTLENECK) == full and sum(1 for o in fused.ops if o.kind == lib.OP_CONV1DW) == part
    m.fuse_bottlenecks = "auto"                    # default rule without a measurement: k <= 5 only
    auto = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"))
    assert len(auto.ops) == nops - 2 * {"n": 4, "s": 4, "m": 2}[scale]
    assert len(Plan(m, 2, 64, 64, lib.F32, lib.F32, torch.device("cpu")).ops) == nops32      # fp32 parity mode never fuses
    m.fuse_head = "auto"                           # head tail: per level 1 depth-wise + 4 convs -> 1 + 1 launches, and no decode launch
    ht = Plan(m, 2, 64, 64, lib.F16, lib.F16, torch.device("cpu"), fuse=False)
    widths = [o.Cin for o in plan.ops if o.kind == lib.OP_CONV1X1 and o.out_f32][::2]
    assert len(widths) == 3
    nf = sum(1 for w in widths if w in (64, 128, 192, 256))         # levels that take the fused tail at this size (256-wide ones only while small): n 3, s 3, m 1
    assert nf == {"n": 3, "s": 3, "m": 1}[scale]
    assert len(ht.ops) == nops - 3 * nf - (1 if nf == 3 else 0)
    assert sum(1 for o in ht.ops if o.kind == lib.OP_HEADTAIL) == nf and any(o.kind == lib.OP_DECODE for o in ht.ops) == (nf < 3)
    for o in ht.ops:
        if o.kind == lib.OP_HEADTAIL:
            assert o.nsrc == 2 and o.src[0].C == o.Cin and o.Win == ht.A and o.w and o.aux[0]
        if o.kind == lib.OP_DECODE:
            assert [bool(o.src[l].ptr) for l in range(3)] == [w not in (64, 128, 192, 256) for w in widths]
    assert [o.Hin for o in ht.ops if o.kind == lib.OP_HEADTAIL] == [0, 64, 80][:nf]
    assert len(Plan(m, 2, 64, 64, lib.F32, lib.F32, torch.device("cpu")).ops) == nops32
    m.fuse_stem = True                             # backbone.0 + backbone.1 in one launch (scale n: 24 -> 48 channels)
    st = Plan(m, 2, 64, 64, lib.F16, lib.U8, torch.device("cpu"), fuse=False)
    if scale in ("n", "s", "m"):                   # ... and the 1x1 that opens backbone.2 rides along (its output goes straight into the concat buffer); m (48 -> 96): round 6
        assert len(st.ops) == len(ht.ops) - 2 and st.ops[0].kind == lib.OP_STEM2
        assert (st.ops[0].ksize, st.ops[0].Cout, st.ops[0].nc, st.ops[0].H, st.ops[0].Hin) == {"n": (24, 48, 48, 16, 64), "s": (32, 64, 64, 16, 64), "m": (48, 96, 96, 16, 64)}[scale]
        # ... as two dense halves (one tensor per slot of RepHDW's concatenation; its closing 1x1 reads them as separate sources)
        c_ = {"n": 24, "s": 32, "m": 48}[scale]
        assert st.ops[0].out_stride == c_ and st.ops[0].aux[0] and st.ops[0].reg_stride == c_ and st.op_names[0] == "backbone.0+1+2.conv1"
        cv2 = st.ops[st.op_names.index("backbone.2.conv2")]
        assert cv2.nsrc == {"n": 3, "s": 4, "m": 4}[scale] and all(cv2.src[k].C == c_ and cv2.src[k].stride == c_ and cv2.src[k].coff == 0 for k in range(cv2.nsrc))
        m.split_cat = False                        # the interleaved concat buffer
        il = Plan(m, 2, 64, 64, lib.F16, lib.U8, torch.device("cpu"), fuse=False)
        assert il.ops[0].out_stride == {"n": 72, "s": 128, "m": 192}[scale] and not il.ops[0].aux[0] and il.ops[il.op_names.index("backbone.2.conv2")].nsrc == 1
        del m.split_cat
        m.fuse_stem = 1
        st1 = Plan(m, 2, 64, 64, lib.F16, lib.U8, torch.device("cpu"), fuse=False)
        assert len(st1.ops) == len(ht.ops) - 1 and st1.ops[0].kind == lib.OP_STEM2 and st1.ops[0].nc == 0 and st1.ops[0].out_stride == st1.ops[0].Cout
        m.fuse_stem = True
    else:
        assert len(st.ops) == len(ht.ops) and st.ops[0].kind == lib.OP_STEM
    m.fuse_mprep = True                            # MPRep's two branches (MaxPool2d + 1x1 | 3x3 s2) in one launch where the kernel exists: 48 / 64 / 96 channels = node 3 of n / s / m (and node 5 of n on bigger batches)
    assert len(Plan(m, 2, 64, 64, lib.F16, lib.U8, torch.device("cpu"), fuse=False).ops) == len(st.ops)     # big maps only (>= 65536 output pixels)
    mp = Plan(m, 32, 384, 384, lib.F16, lib.U8, torch.device("cpu"), fuse=False)
    m.fuse_mprep = False
    st = Plan(m, 32, 384, 384, lib.F16, lib.U8, torch.device("cpu"), fuse=False)
    m.fuse_mprep = True
    assert len(mp.ops) == len(st.ops) - 1                 # node 3 of every scale at this size: 48 / 64 channels on the LDS-resident 3x3 kernel, 96 (m) on the register-resident one
    if True:
        o = mp.ops[mp.op_names.index("backbone.3.conv1+conv2")]
        c = {"n": 48, "s": 64, "m": 96}[scale]
        assert (o.kind, o.tile_k, o.nc, o.reg_stride, o.out_coff, o.Cin, o.Cout, o.act) == (lib.OP_CONV3X3S2, 7 if c == 96 else 6, c, 0, c, c, c, lib.ACT_RELU)
        assert "backbone.3.conv1" not in mp.op_names and "backbone.3.conv2" not in mp.op_names
    assert len(Plan(m, 2, 64, 64, lib.F32, lib.F32, torch.device("cpu")).ops) == nops32      # the fp32 parity plan keeps them apart


def test_product_synth_generator_equals_oracle_generator():
    from maf_yolo_amd import synth
    for s in "nsm":
        a = synth.synth_state_dict(M.Model(s), s, 0)
        b = O.synth_state_dict(s, 0)
        assert list(a.keys()) == list(b.keys())
        assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(synth.synth_images(2, 32, 5), O.synth_images(2, 32, 5))


def test_tune_cache_roundtrip(tmp_path):
    from maf_yolo_amd import engine
    saved = dict(engine._TUNE_CACHE)
    try:
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE[(1, 0, 204800, 48, 48, 1, (0,), 0)] = (2, 4, 1)
        engine._TUNE_CACHE[(3, 0, 32, 80, 80, 96, 5, 2)] = (8, 16, 32)
        p = str(tmp_path / "tune.json")
        engine.save_tune_cache(p)
        want = dict(engine._TUNE_CACHE)
        engine._TUNE_CACHE.clear()
        assert engine.load_tune_cache(p) == 2
        assert engine._TUNE_CACHE == want
    finally:
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE.update(saved)


def test_checkpoint_bridge_reads_reference_pt_without_reference_code(golden):
    """f3: a checkpoint pickled by the reference's trainer (modules, fp16, 'model' + 'ema') loads with no yolov6 on sys.path."""
    import hashlib
    import sys
    from maf_yolo_amd import checkpoint
    assert "yolov6" not in sys.modules
    g = golden("ref_ckpt_tiny")
    path = os.path.join(os.path.dirname(__file__), "golden", "ref_ckpt_tiny.pt")
    sd, yaml_dict, nc, raw = checkpoint.read_reference_checkpoint(path)
    assert raw["epoch"] == 3 and raw["updates"] == 17 and nc == 80 and yaml_dict["width_multiple"] == 0.125
    m = checkpoint.load_checkpoint(path)
    msd = m.state_dict()
    assert len(msd) == int(g["n_keys"]) and not m.training and all(v.dtype != torch.float16 for v in msd.values())
    h = int(hashlib.sha1("\n".join("%s %s" % (k, tuple(v.shape)) for k, v in msd.items()).encode()).hexdigest()[:12], 16)
    assert h == int(g["key_hash"][0])                                     # same names, same order, same shapes as the reference's
    assert abs(float(sum(v.double().sum() for v in msd.values())) - float(g["checksum"][0])) < 1e-6 * abs(float(g["checksum"][0])) + 1e-6   # the EMA weights
    back = checkpoint.reference_state_dict(m)
    assert list(back) == list(msd)


def test_fused_kernel_records_match_the_c_abi_sizes(built):
    """pack_head_tail / pack_stem2 produce exactly the bytes the C side declares, with the documented fragment layout."""
    g = torch.Generator().manual_seed(5)
    for C in (64, 128, 192):
        w1, b1 = torch.randn(C, C, 1, 1, generator=g), torch.randn(C, generator=g)
        w2, b2 = torch.randn(68, C, 1, 1, generator=g), torch.randn(68, generator=g)
        rec = pack.pack_head_tail(w1, b1, w2, b2)
        assert rec.numel() == built.maf_head_tail_record_bytes(C)
        f1 = rec[:C * C * 2].view(torch.float16).view(C // 16, C // 32, 4, 16, 8)            # [t][ks][g][i][j]
        t, ks, gg, i, j = 1, C // 32 - 1, 3, 5, 6
        assert f1[t, ks, gg, i, j] == w1[16 * t + i, 32 * ks + 8 * gg + j, 0, 0].half()
        f2 = rec[C * C * 2:C * C * 2 + 80 * C * 2].view(torch.float16).view(5, C // 32, 4, 16, 8)   # [t2][j][g][n][q]
        t2, jj, gg, n, q = 4, 1, 2, 3, 5                                                       # column 67: the last real DFL logit
        assert f2[t2, jj, gg, n, q] == w2[16 * t2 + n, 32 * jj + 16 + 4 * gg + q - 4, 0, 0].half()
        assert torch.all(f2[4, :, :, 4:, :] == 0)                                             # columns 68..79: zero padding
        tail = rec[C * C * 2 + 80 * C * 2:].view(torch.float32)
        assert torch.equal(tail[:C], b1) and torch.equal(tail[C:C + 68], b2) and torch.all(tail[C + 68:] == 0)
    for c0, c1 in ((24, 48), (32, 64)):
        w0, b0 = torch.randn(c0, 3, 3, 3, generator=g), torch.randn(c0, generator=g)
        w1, b1 = torch.randn(c1, c0, 3, 3, generator=g), torch.randn(c1, generator=g)
        rec = pack.pack_stem2(w0, b0, w1, b1)
        assert rec.numel() == built.maf_stem2_record_bytes(c0, c1, 0)
        w3, b3 = torch.randn(c1, c1, 1, 1, generator=g), torch.randn(c1, generator=g)
        rec3 = pack.pack_stem2(w0, b0, w1, b1, w3, b3)
        assert rec3.numel() == built.maf_stem2_record_bytes(c0, c1, c1) and torch.equal(rec3[:rec.numel()], rec)
        ks3 = (c1 + 31) // 32
        f3 = rec3[rec.numel():rec.numel() + ks3 * (c1 // 16) * 1024].view(torch.float16).view(c1 // 16, ks3, 4, 16, 8)      # [t][j][g][i][q]
        assert f3[1, 0, 2, 5, 6] == w3[16 + 5, 16 + 4 * 2 + 6 - 4, 0, 0].half() and f3[2, 1, 3, 0, 1] == w3[32, 32 + 12 + 1, 0, 0].half()
        if c1 == 48:
            assert torch.all(f3[:, 1, :, :, 4:] == 0)                                              # channels 48..63 do not exist
        assert torch.equal(rec3[-c1 * 4:].view(torch.float32), b3)
        f0 = rec[:2048].view(torch.float16).view(2, 4, 16, 8)                                  # [t][g][n][j]
        assert f0[1, 2, 3, 4] == w0[19, 2, 0, 2].half()                                        # k = 8*2+4 = 20 = (c 2, ky 0, kx 2), channel 16+3
        assert torch.all(f0[:, 3, :, 3:] == 0)                                                 # taps 27..31 do not exist
        gr = c0 // 8
        ks1 = (9 * gr + 3) // 4
        f1 = rec[2048:2048 + ks1 * (c1 // 16) * 1024].view(torch.float16).view(ks1, c1 // 16, 4, 16, 8)   # [s][t][g][n][j]
        s_, t, gg, n, j = 2, 1, 1, 7, 3
        tap, grp = divmod(4 * s_ + gg, gr)
        assert f1[s_, t, gg, n, j] == w1[16 * t + n, 8 * grp + j, tap // 3, tap % 3].half()


def test_training_loss_has_no_cpu_path():
    crit = M.ComputeLoss()
    assert crit.warmup_epoch == 3                       # the reference's default (yolov6/models/loss.py:23), which its trainer relies on
    feats = [torch.zeros(1, 8, s, s) for s in (8, 4, 2)]
    with pytest.raises(lib.MafError):
        crit((feats, torch.rand(1, 84, 80), torch.randn(1, 84, 68)), torch.zeros(0, 6), 0, 0)


def test_checkpoint_unpickler_runs_nothing_from_a_crafted_file(tmp_path):
    """only an explicit allow-list of globals is resolved; builtins.eval / exec / getattr, os.system, functools.partial,
    torch.hub loaders ... named by a pickle become inert stand-ins (constructed, never executed)."""
    import pickle
    from maf_yolo_amd import checkpoint
    marker = tmp_path / "pwned"

    class Evil:
        def __init__(self, fn, args):
            self.fn, self.args = fn, args

        def __reduce__(self):
            return self.fn, self.args

    import functools
    import os as _os
    payloads = [Evil(eval, ("open(%r, 'w').write('x')" % str(marker),)), Evil(_os.system, ("touch %s" % marker,)),
                Evil(functools.partial, (_os.system, "touch %s" % marker)), Evil(getattr, (dict, "fromkeys")),
                Evil(exec, ("import os; os.system('touch %s')" % marker,)), Evil(__import__, ("subprocess",))]
    for p in payloads:
        blob = pickle.dumps({"model": p, "epoch": 1}, protocol=2)
        out = checkpoint._PickleModule.loads(blob)
        assert not marker.exists()
        assert isinstance(out["model"], checkpoint._Inert) and out["epoch"] == 1
    for mod, name in (("builtins", "eval"), ("builtins", "exec"), ("builtins", "getattr"), ("builtins", "__import__"), ("functools", "partial"),
                      ("torch.hub", "load"), ("torch.utils.cpp_extension", "load"), ("os", "system"), ("torch", "load"), ("numpy", "load")):
        assert not checkpoint._allowed(mod, name), (mod, name)
    for mod, name in (("torch._utils", "_rebuild_tensor_v2"), ("torch", "HalfStorage"), ("torch", "float16"), ("collections", "OrderedDict")):
        assert checkpoint._allowed(mod, name), (mod, name)


def test_plan_cache_fingerprint_tracks_in_place_weight_updates():
    """cached eval plans hold packed weight copies; the reference evaluates its EMA model, which `ema.update` mutates in
    place every step without ever calling train() (engine.py:246).  The fingerprint must move on every in-place update."""
    m = M.Model("n").eval()
    v0 = m.weights_version()
    assert m.weights_version() == v0
    with torch.no_grad():
        for p in m.parameters():                       # what ModelEMA.update does: v *= d; v += (1 - d) * msd[k]
            p.mul_(0.999)
            break
    v1 = m.weights_version()
    assert v1 != v0
    m.backbone[0].rbr_dense.bn.running_mean.add_(1.0)   # buffers count too
    assert m.weights_version() != v1
    m._plans["sentinel"] = object(); m._plans_version = m.weights_version()
    m.float()                                           # _apply-style conversions drop the plans
    assert m._plans == {} and m._plans_version is None
    m._plans["sentinel"] = object()
    m.load_state_dict(m.state_dict())
    assert m._plans == {}
    import copy
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        next(m2.parameters()).add_(1.0)
    assert m2.weights_version() != m.weights_version()  # the copy tracks its own tensors


def test_integration_md_ctypes_stub_is_the_c_struct(built):
    """VERDICT r1 / ADVICE r1: the MafOp stub printed in INTEGRATION.md must be maf_op_t field for field (a short stub makes
    maf_engine_create read past every op).  The stub is parsed out of the document and compared with lib.MafOp and the library's own
    sizeof(maf_op_t)."""
    import ctypes as C
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    m = re.search(r"class MafSrc\(C\.Structure\):.*?class MafOp\(C\.Structure\):.*?\n\n", text, re.S)
    assert m, "stub not found in INTEGRATION.md"
    src = re.sub(r"^assert L\..*$", "", m.group(0), flags=re.M)
    ns = {"C": C}
    exec(src, ns)                                           # the document's own text (ours, not the reference's)
    doc_op, doc_src = ns["MafOp"], ns["MafSrc"]
    assert [(n, C.sizeof(t)) for n, t in doc_op._fields_] == [(n, C.sizeof(t)) for n, t in lib.MafOp._fields_]
    assert [(n, C.sizeof(t)) for n, t in doc_src._fields_] == [(n, C.sizeof(t)) for n, t in lib.MafSrc._fields_]
    assert C.sizeof(doc_op) == C.sizeof(lib.MafOp) == built.maf_op_size()
    for n, _ in lib.MafOp._fields_:
        assert getattr(doc_op, n).offset == getattr(lib.MafOp, n).offset, n


def test_torch_custom_op_library_loads_and_declares_the_ops(built):
    """SURVEY.md 8(b) / north_star: the hot path as PyTorch custom ops — torch.ops.load_library works, every op has its schema, and the fake
    (meta) kernels trace without a device (no compute here)."""
    from maf_yolo_amd import torch_ops
    ops = torch_ops.load()
    for name in torch_ops.OPS:
        assert hasattr(ops, name), name
    assert str(torch.ops.mafyolo.decode_nms.default._schema) == ("mafyolo::decode_nms(Tensor pred, float conf_thres, float iou_thres, bool agnostic, "
                                                                  "bool multi_label, int max_det, int[]? classes) -> (Tensor, Tensor)")
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 64, 9, 12, device="cuda", dtype=torch.float16)
        y = ops.conv3x3s2_bias_act(x, torch.empty(32, 64, 3, 3, device="cuda"), None, 0)
        assert y.shape == (2, 32, 5, 6) and y.dtype == torch.float16 and y.is_contiguous(memory_format=torch.channels_last)
        assert ops.conv1x1_bias_act(x, torch.empty(80, 64, 1, 1, device="cuda"), torch.empty(80, device="cuda"), 3).shape == (2, 80, 9, 12)
        p = ops.head_decode([torch.empty(2, 80, s, s, device="cuda") for s in (8, 4, 2)], [torch.empty(2, 68, s, s, device="cuda") for s in (8, 4, 2)], [8.0, 16.0, 32.0])
        assert p.shape == (2, 84, 85) and p.dtype == torch.float32
    with pytest.raises(Exception):                           # a CPU tensor has no kernel: there is no CPU fallback
        ops.conv1x1_bias_act(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 1, 1), None, 0)


def test_model_dispatch_ops_traces_as_one_graph_without_a_device(built):
    """VERDICT r2 task 8: `Model(dispatch="ops").traceable()` is ONE dynamo graph (fullgraph=True: no break) of torch.ops.mafyolo.* calls + cat / upsample,
    traced through the fake kernels alone (meta tensors: nothing is computed here)."""
    import maf_yolo_amd as M
    m = M.Model("n", dispatch="ops").eval()
    stride = m.detect.stride.clone()
    m = m.to("meta")
    m.detect.stride = stride
    f = m.traceable(torch.float32)
    seen = []

    def backend(gm, example_inputs):
        seen.append([str(nd.target) for nd in gm.graph.nodes if nd.op == "call_function"])
        return gm.forward
    y = torch.compile(f, backend=backend, fullgraph=True)(torch.empty(2, 3, 64, 64, device="meta").contiguous(memory_format=torch.channels_last))
    assert y.shape == (2, 84, 85) and len(seen) == 1
    ours = [t for t in seen[0] if "mafyolo" in t]
    assert len(ours) == 85 and {t.split(".")[1] for t in ours} == {"conv3x3s2_bias_act", "conv1x1_bias_act", "dwconv_bias_act", "mprep", "sppf", "head_decode"}
    rest = {t for t in seen[0] if "mafyolo" not in t}
    assert all(("cat" in t) or ("interpolate" in t) or ("getitem" in t) for t in rest), rest


@pytest.mark.parametrize("k", [3, 5, 7, 9])
def test_weight_pairs_and_pixel_pairs_reproduce_a_depthwise_row(k):
    """pack.pack_dw_pairs / pack.pairs_from_nhwc against the index algebra of csrc/dwconv_p2.hip, evaluated on the host: output pixel t of a 4-pixel
    strip takes, per kernel row, the pixel pairs j = m + ((t + e) >> 1) of its window (which starts PE = P + e pixels left of the strip, e = P & 1)
    with the EVEN weight pairs if t + e is even and the ODD ones otherwise — (k + 1) / 2 two-tap products per row instead of k taps.  Must equal
    the plain depth-wise convolution (F.conv2d) for every kernel size, both window alignments, all 8 channels of a group."""
    g = torch.Generator().manual_seed(40 + k)
    C_, H, W = 16, 6, 12
    x = torch.randn(1, C_, H, W, generator=g).half().float()
    w = (torch.randn(C_, 1, k, k, generator=g) / k).half().float()
    ref = torch.nn.functional.conv2d(x, w, None, 1, k // 2, 1, C_)[0]                    # [C, H, W]
    wp = pack.pack_dw_pairs(w).float()                                                   # [C/8, k, 2 (h), 2 (set), NP, 4 (d), 2]
    xp = pack.pairs_from_nhwc(x.permute(0, 2, 3, 1).contiguous())[0].float()             # [H, W/2, C, 2]
    P = k // 2
    e = P & 1
    PE, NP = P + e, (k + 1) // 2
    assert wp.shape == (C_ // 8, k, 2, 2, NP, 4, 2)

    def pair(y, q, c):                                                                   # the zero page outside the image
        return xp[y, q, c] if 0 <= y < H and 0 <= q < W // 2 else torch.zeros(2)
    for cg in range(C_ // 8):
        for y in range(H):
            for xs in range(0, W, 4):                                                    # a lane's strip
                q0 = (xs - PE) // 2                                                      # first pair of its window
                for t in range(4):
                    for h in range(2):
                        for d in range(4):
                            c = cg * 8 + 4 * h + d
                            acc = 0.0
                            for ky in range(k):
                                for m in range(NP):
                                    px = pair(y + ky - P, q0 + m + ((t + e) >> 1), c)
                                    wq = wp[cg, ky, h, (t + e) & 1, m, d]
                                    acc += float(px[0] * wq[0] + px[1] * wq[1])
                            assert abs(acc - float(ref[c, y, xs + t])) < 1e-4, (k, c, y, xs + t)


def test_build_optimizer_groups_match_the_reference_counts():
    """yolov6/solver/build.py:12-33: BatchNorm weights (no decay), other weights (decay), biases (no decay).  The expected (count, elements) per
    group were read off the reference's own build_optimizer on its n / s / m models in the build container (same parameter names, same order)."""
    import importlib
    M = importlib.import_module("maf-yolo_amd")
    want = {"n": [(140, 25088), (131, 3964745), (146, 25532)], "s": [(203, 54336), (184, 9012625), (209, 54780)],
            "m": [(265, 118272), (236, 24699857), (271, 118716)]}
    for scale, groups in want.items():
        opt = M.build_optimizer(M.Model(scale), lr0=0.02, momentum=0.9, weight_decay=5e-4)
        got = [(len(g["params"]), sum(p.numel() for p in g["params"])) for g in opt.param_groups]
        assert got == groups, (scale, got)
        assert [g["weight_decay"] for g in opt.param_groups] == [0, 5e-4, 0]
        assert all(g["nesterov"] and g["momentum"] == 0.9 and g["lr"] == 0.02 for g in opt.param_groups)
        assert not opt.param_groups[0].get("fused")            # CPU parameters: the plain implementation


def test_model_ema_matches_the_per_tensor_rule_bit_for_bit():
    """yolov6/utils/ema.py:29-37: every floating state_dict entry e <- e * d + (1 - d) * m with d = decay * (1 - exp(-updates / 2000)),
    integer buffers (num_batches_tracked) untouched; the multi-tensor update must round exactly like the per-tensor loop."""
    import copy
    import importlib
    import math
    M = importlib.import_module("maf-yolo_amd")
    torch.manual_seed(3)
    model = M.Model("n")
    ema = M.ModelEMA(model)
    want = copy.deepcopy(model.state_dict())
    assert not ema.ema.training and all(not p.requires_grad for p in ema.ema.parameters())
    for step in range(1, 4):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.01)
            for name, b in model.named_buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.1)
                else:
                    b.add_(1)
        ema.update(model)
        d = 0.9999 * (1 - math.exp(-step / 2000))
        sd = model.state_dict()
        for k, v in want.items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1 - d) * sd[k]
    assert ema.updates == 3
    got = ema.ema.state_dict()
    assert list(got) == list(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert int(got["backbone.0.rbr_dense.bn.num_batches_tracked"]) == 0 and int(model.state_dict()["backbone.0.rbr_dense.bn.num_batches_tracked"]) == 3
    model.nc, model.names = 3, ["a", "b", "c"]
    ema.update_attr(model, include=["nc", "names", "stride"])
    assert ema.ema.nc == 3 and ema.ema.names == ["a", "b", "c"]


def test_model_ema_follows_replaced_parameter_storage():
    """ModelEMA caches the (average, model) tensor pairs; when a parameter's storage is replaced after the first update (`p.data = ...`,
    `.to()`), the next update must read the NEW storage — the reference re-reads state_dict() every time (yolov6/utils/ema.py:29-37)."""
    import importlib
    import math
    M = importlib.import_module("maf-yolo_amd")
    torch.manual_seed(4)
    model = M.Model("n")
    ema = M.ModelEMA(model)
    ema.update(model)
    p = model.backbone[0].rbr_dense.conv.weight
    before = ema.ema.backbone[0].rbr_dense.conv.weight.detach().clone()
    p.data = torch.full_like(p, 5.0)                                   # new storage, the cached source tensor is dead
    ema.update(model)
    d = 0.9999 * (1 - math.exp(-2 / 2000))
    want = before * d + (1 - d) * 5.0
    assert torch.allclose(ema.ema.backbone[0].rbr_dense.conv.weight, want, rtol=0, atol=1e-7)


def test_model_ema_follows_buffers_replaced_by_apply():
    """Round 4: ModelEMA validates its cached tensor lists with the model's `_apply` generation + the addresses of the cached Parameter objects (the
    full walk of both module trees cost 3-4 ms of host time per step).  `_apply` (.to() / .double() / .float() ...) REPLACES buffer tensors: the running
    statistics the average reads afterwards must be the new ones."""
    import importlib
    import math
    M = importlib.import_module("maf-yolo_amd")
    torch.manual_seed(5)
    model = M.Model("n")
    ema = M.ModelEMA(model)
    ema.update(model)
    g0 = model._maf_apply_gen if hasattr(model, "_maf_apply_gen") else 0
    model.double().float()                                             # every buffer and every parameter's storage is a new tensor now
    assert model._maf_apply_gen >= g0 + 2
    key = "backbone.8.conv2.bn.running_mean"
    before = ema.ema.state_dict()[key].clone()
    with torch.no_grad():
        model.state_dict()[key].fill_(3.0)
    ema.update(model)
    d = 0.9999 * (1 - math.exp(-2 / 2000))
    assert torch.allclose(ema.ema.state_dict()[key], before * d + (1 - d) * 3.0, rtol=0, atol=1e-7)
    sig = ema._sig
    ema.update(model)
    assert ema._sig == sig and sig[0] == "gen"                         # steady state: the cheap signature, unchanged


def test_cat_buffer_join_and_fork_are_cat_and_split_for_autograd():
    """train_ops.CatBuffer / join / fork (the copy-free concats of the train-form graph): with parts that no producer stored into their slots `join` copies them in —
    it IS torch.cat, forward and backward; a part that already sits in its slot is not copied; `fork` returns (t, t[:, lo:]) and adds the tail's gradient into
    the gradient of t (what autograd's sum of the two consumers gives)."""
    import importlib
    to = importlib.import_module("maf-yolo_amd.train_ops")
    torch.manual_seed(0)
    a = torch.randn(2, 8, 3, 5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(2, 16, 3, 5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(2, 24, 3, 5)
    cb = to.CatBuffer(a, [8, 16])
    n0 = to.stats.get("cat_copied_parts", 0)
    a2, tail = to.fork(a * 2.0, 4)
    y = to.join(cb, [a2, b])
    assert to.stats.get("cat_copied_parts", 0) - n0 == 2
    ((y * w).sum() + (tail * 3.0).sum()).backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    t = a * 2.0
    y2 = torch.cat([t, b], 1)
    ((y2 * w).sum() + (t[:, 4:] * 3.0).sum()).backward()
    assert torch.equal(y.detach(), y2.detach())
    assert torch.allclose(ga, a.grad) and torch.allclose(gb, b.grad)
    # a resident part: written into its slot beforehand, handed to join as the slot tensor itself
    cb2 = to.CatBuffer(a, [8, 16])
    s0 = cb2.slot(0)
    s0.copy_(a.detach())
    assert not s0._is_view() and s0.data_ptr() == cb2.buf.data_ptr()
    n0 = to.stats.get("cat_copied_parts", 0)
    y3 = to.join(cb2, [s0, b])
    assert to.stats.get("cat_copied_parts", 0) - n0 == 1
    assert torch.equal(y3.detach(), torch.cat([a, b], 1).detach())
    # mixed dtypes: the framework's promotion rule, through torch.cat
    assert to.join(to.CatBuffer(a, [8, 16]), [a.detach(), b.detach().double()]).dtype == torch.float64


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver types it for N = 1 — no launcher, WORLD_SIZE unset — re-executes itself under torch.distributed.run
    (bench.self_launch), forms a two-rank group and runs the timing protocol (barrier, MAX over ranks, one JSON line from rank 0).  `--rendezvous-only` keeps the
    device out of it (this container has none); the same launch with kernels is tests/test_gpu_train.py::test_bench_train_self_launch_two_ranks_on_one_device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--rendezvous-only"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["world_size"] == 2 and d["launched_by"] == "self_launch"
    assert d["max_over_ranks_s"] >= 0.019 > d["rank0_s"]              # rank 1 slept 20 ms, rank 0 10 ms: the line carries the MAX
    # a launcher that started the wrong number of ranks is refused, not silently benchmarked
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--rendezvous-only"], cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
