"""CPU oracle for the MAF-YOLO hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (maf-yolo_amd/) never imports it and fails loudly when its HIP library is
missing.

What it is: an independent plain-PyTorch-fp32 / NumPy restatement of the reference algorithm
for the path BASELINE.json's north_star names, each function citing the reference file:line it
follows (paths relative to /root/reference):

  * architecture tables for MAF-YOLO-n/s/m      configs/yaml/MAF-YOLO-{n,s,m}.yaml,
                                                 yolov6/models/yolo.py:15-120 (parse_model)
  * train-form forward (eval-mode BN)            yolov6/layers/common.py (classes cited below)
  * deploy switch (re-parameterisation algebra)  yolov6/utils/torch_utils.py:50-98,
                                                 common.py:226-283, 2636-2645, 2926-2947,
                                                 3033-3051, 3085-3100
  * deploy-form forward                          same classes, `forward_fuse` / `rbr_reparam`
  * DFL decode                                   yolov6/models/yolo.py:355-396,
                                                 yolov6/assigners/anchor_generator.py:11-25,
                                                 yolov6/utils/general.py:29-40
  * non_max_suppression                          yolov6/utils/nms.py:21-105

Pinning: tests/test_oracle_golden.py checks every function here against fixtures under
tests/golden/ that tools/make_golden.py generated in the build container by importing the
reference itself (SURVEY.md Appendix C recipe).  ONE boundary stays *parity unpinned*: the inner
greedy NMS, which the reference takes from `torchvision.ops.nms` (nms.py:96) — torchvision is
absent from this image and no reference test covers it, so `greedy_nms` below restates
torchvision's documented CPU algorithm (stable descending sort; suppress j when
IoU(i,j) > iou_threshold, threshold compared in double) and the reference's surrounding logic
was run with this function stubbed in.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # yolov6/utils/torch_utils.py:43-45 (initialize_weights overrides torch's 1e-5)
REG_MAX = 16
NUM_CLASSES = 80
STRIDES = (8, 16, 32)


# --------------------------------------------------------------------------------------------
# Architecture tables (configs/yaml/MAF-YOLO-{n,s,m}.yaml after the width rules of
# yolov6/models/yolo.py:28-32,56-59,92-96)
# --------------------------------------------------------------------------------------------
def _mkdiv(x, d):
    return int(math.ceil(x / d) * d)


_SCALE = {
    #        gw     backbone RepHDW (cout, depth)             neck literals
    "n": dict(gw=0.375, bb=[(48, 1), (96, 1), (192, 1), (384, 1)],
              cw10=96, h12=(192, 1), cw14=64, h16=(128, 1), cw18=64, h20=(128, 1), h22=(128, 1),
              cw23=128, h26=(128, 1), cw27=128, h30=(192, 1), heads=(341, 341, 512)),
    "s": dict(gw=0.5, bb=[(64, 2), (128, 2), (256, 2), (512, 2)],
              cw10=128, h12=(256, 2), cw14=96, h16=(192, 2), cw18=96, h20=(192, 2), h22=(192, 2),
              cw23=192, h26=(192, 2), cw27=192, h30=(256, 2), heads=(384, 384, 512)),
    "m": dict(gw=0.75, bb=[(96, 2), (192, 4), (384, 4), (768, 2)],
              cw10=256, h12=(512, 3), cw14=192, h16=(384, 3), cw18=192, h20=(384, 3), h22=(256, 3),
              cw23=192, h26=(384, 3), cw27=192, h30=(384, 3), heads=(341, 512, 512)),
}


def arch(scale):
    """Node list [(index, from, op, args)] in YAML order; `from` as in the YAML (-1 = previous)."""
    s = _SCALE[scale]
    gw = s["gw"]
    n = []
    n.append((-1, "repvgg", dict(cout=_mkdiv(64 * gw, 4))))
    n.append((-1, "repvgg", dict(cout=_mkdiv(128 * gw, 4))))
    n.append((-1, "rephdw", dict(cout=s["bb"][0][0], depth=s["bb"][0][1], k=3)))
    n.append((-1, "mprep", dict(cout=_mkdiv(256 * gw, 8))))
    n.append((-1, "rephdw", dict(cout=s["bb"][1][0], depth=s["bb"][1][1], k=5)))
    n.append((-1, "mprep", dict(cout=_mkdiv(512 * gw, 8))))
    n.append((-1, "rephdw", dict(cout=s["bb"][2][0], depth=s["bb"][2][1], k=7)))
    n.append((-1, "mprep", dict(cout=_mkdiv(1024 * gw, 8))))
    n.append((-1, "rephdw", dict(cout=s["bb"][3][0], depth=s["bb"][3][1], k=9)))
    n.append((-1, "sppf", dict(cout=_mkdiv(1024 * gw, 4))))
    n.append((6, "cw", dict(cout=s["cw10"])))                                   # 10
    n.append(([-1, 9], "concat", {}))                                           # 11
    n.append((-1, "rephdw", dict(cout=s["h12"][0], depth=s["h12"][1], k=9)))    # 12
    n.append((-1, "up", {}))                                                    # 13
    n.append((4, "cw", dict(cout=s["cw14"])))                                   # 14
    n.append(([-1, 6, -2], "concat", {}))                                       # 15
    n.append((-1, "rephdw", dict(cout=s["h16"][0], depth=s["h16"][1], k=7)))    # 16
    n.append((-1, "up", {}))                                                    # 17
    n.append((2, "cw", dict(cout=s["cw18"])))                                   # 18
    n.append(([-1, 4, -2], "concat", {}))                                       # 19
    n.append((-1, "rephdw", dict(cout=s["h20"][0], depth=s["h20"][1], k=5)))    # 20
    n.append(([-1, 17], "concat", {}))                                          # 21
    n.append((-1, "rephdw", dict(cout=s["h22"][0], depth=s["h22"][1], k=5)))    # 22  P3
    n.append((-1, "cw", dict(cout=s["cw23"])))                                  # 23
    n.append((20, "cw", dict(cout=s["cw23"])))                                  # 24
    n.append(([-2, -1, 16, 13], "concat", {}))                                  # 25
    n.append((-1, "rephdw", dict(cout=s["h26"][0], depth=s["h26"][1], k=7)))    # 26  P4
    n.append((-1, "cw", dict(cout=s["cw27"])))                                  # 27
    n.append((16, "cw", dict(cout=s["cw27"])))                                  # 28
    n.append(([-2, -1, 12], "concat", {}))                                      # 29
    n.append((-1, "rephdw", dict(cout=s["h30"][0], depth=s["h30"][1], k=9)))    # 30  P5
    n.append((22, "head", dict(ch=_mkdiv(s["heads"][0] * gw, 8), k=5)))         # 31
    n.append((26, "head", dict(ch=_mkdiv(s["heads"][1] * gw, 8), k=7)))         # 32
    n.append((30, "head", dict(ch=_mkdiv(s["heads"][2] * gw, 8), k=9)))         # 33
    return [(i, f, op, a) for i, (f, op, a) in enumerate(n)]


def dil_branches(k):
    """DilatedReparamBlock branch kernel sizes, all dilation 1 (common.py:2997-3008)."""
    return {9: (7, 5, 3), 7: (5, 3), 5: (3, 1), 3: (3, 1)}[k]


# --------------------------------------------------------------------------------------------
# state_dict layout (the reference's key names; SURVEY.md §7 "state_dict compatibility")
# --------------------------------------------------------------------------------------------
def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def _conv_keys(prefix, cin, cout, k):     # Conv = conv + bn (common.py:29-50)
    return [(prefix + ".conv.weight", (cout, cin, k, k))] + _bn_keys(prefix + ".bn", cout)


def _unirep_keys(prefix, c, k):           # UniRepLKNetBlock (common.py:3053-3083) + DilatedReparamBlock
    keys = [(prefix + ".dwconv.lk_origin.weight", (c, 1, k, k))] + _bn_keys(prefix + ".dwconv.origin_bn", c)
    for kk in dil_branches(k):
        keys.append((prefix + ".dwconv.dil_conv_k%d_1.weight" % kk, (c, 1, kk, kk)))
        keys += _bn_keys(prefix + ".dwconv.dil_bn_k%d_1" % kk, c)
    return keys + _bn_keys(prefix + ".norm", c)


def _repvgg_keys(prefix, cin, cout):      # RepVGGBlock stride 2: no identity branch (common.py:210)
    return (_conv_keys(prefix + ".rbr_dense", cin, cout, 3) + _conv_keys(prefix + ".rbr_1x1", cin, cout, 1))


def channels_out(scale):
    """Output channels of every node (ch[] of parse_model, yolo.py:117-119)."""
    ch = []
    for i, f, op, a in arch(scale):
        if op == "concat":
            c = sum(ch[i + x if x < 0 else x] for x in f)
        elif op == "up":
            c = ch[i - 1]
        elif op == "head":
            c = a["ch"]
        else:
            c = a["cout"]
        ch.append(c)
    return ch


def _cin(i, f, ch):
    if i == 0:
        return 3
    return ch[i + f if f < 0 else f]


def state_spec(scale):
    """Ordered [(key, shape)] of the train-form state_dict (838 / 1206 / 1568 tensors for n/s/m)."""
    ch = channels_out(scale)
    keys = []
    for i, f, op, a in arch(scale):
        p = "backbone.%d" % i
        if op == "repvgg":
            keys += _repvgg_keys(p, _cin(i, f, ch), a["cout"])
        elif op == "rephdw":
            cin, cout, c_ = _cin(i, f, ch), a["cout"], int(a["cout"] * 0.5)
            keys += _conv_keys(p + ".conv1", cin, 2 * c_, 1)
            for d in range(a["depth"]):
                q = p + ".m.%d" % d
                keys += _conv_keys(q + ".conv1", c_, 3 * c_, 1)
                keys += _unirep_keys(q + ".conv2", 3 * c_, a["k"])
                keys += _conv_keys(q + ".one_conv", 3 * c_, c_, 1)
            keys += _conv_keys(p + ".conv2", c_ * (a["depth"] + 2), cout, 1)
        elif op == "mprep":
            cin, c_ = _cin(i, f, ch), a["cout"] // 2
            keys += _conv_keys(p + ".conv1", cin, c_, 1)
            keys += _repvgg_keys(p + ".conv2", cin, c_)
        elif op == "sppf":
            cin, c_ = _cin(i, f, ch), _cin(i, f, ch) // 2
            keys += _conv_keys(p + ".cv1", cin, c_, 1)
            keys += _conv_keys(p + ".cv2", 4 * c_, a["cout"], 1)
        elif op == "cw":
            keys += _conv_keys(p + ".block", _cin(i, f, ch), a["cout"], 3)
        elif op == "head":
            cin, c = ch[f], a["ch"]
            keys += _conv_keys(p + ".stem", cin, c, 1)
            keys += _unirep_keys(p + ".cls_conv", c, a["k"])
            keys += _conv_keys(p + ".cls_conv_s", c, c, 1)
            keys += _unirep_keys(p + ".reg_conv", c, a["k"])
            keys += _conv_keys(p + ".reg_conv_s", c, c, 1)
            keys += [(p + ".cls_pred.weight", (NUM_CLASSES, c, 1, 1)), (p + ".cls_pred.bias", (NUM_CLASSES,)),
                     (p + ".reg_pred.weight", (4 * (REG_MAX + 1), c, 1, 1)), (p + ".reg_pred.bias", (4 * (REG_MAX + 1),))]
    keys += [("detect.proj", (REG_MAX + 1,)), ("detect.proj_conv.weight", (1, REG_MAX + 1, 1, 1))]
    return keys


_SYNTH_GAIN = {"n": 1.384, "s": 1.438, "m": 1.462}


def synth_state_dict(scale, seed=0, cls_bias=-5.5):
    """Deterministic synthetic train-form weights (NumPy legacy RandomState => identical on every box).

    Conv weights ~ N(0, g^2/fan_in) with a per-scale gain so activations stay O(1) through the net; BN affine and
    running statistics randomised (the reference's defaults would make re-param a no-op);
    `cls_pred/reg_pred` weights randomised and `cls_pred.bias ~ N(cls_bias, 0.5)` because the reference
    zero-initialises them (common.py:1307-1323), which would make outputs input-independent
    (SURVEY.md §8c caveat).
    """
    rs = np.random.RandomState(seed)
    sd = {}
    g_conv = _SYNTH_GAIN[scale]           # calibrated so |activation| stays O(1) through the net (fp16-safe)
    for key, shape in state_spec(scale):
        if key.endswith("num_batches_tracked"):
            v = np.array(0, dtype=np.int64)
        elif key == "detect.proj":
            v = np.linspace(0, REG_MAX, REG_MAX + 1, dtype=np.float32)
        elif key == "detect.proj_conv.weight":
            v = np.linspace(0, REG_MAX, REG_MAX + 1, dtype=np.float32).reshape(shape)
        elif key.endswith("running_mean"):
            v = (rs.randn(*shape) * 0.1).astype(np.float32)
        elif key.endswith("running_var"):
            v = (0.5 + rs.rand(*shape)).astype(np.float32)
        elif ".bn." in key or "_bn" in key or ".norm." in key:
            if key.endswith(".weight"):
                v = (0.7 + 0.6 * rs.rand(*shape)).astype(np.float32)
            else:
                v = (rs.randn(*shape) * 0.1).astype(np.float32)
        elif key.endswith("cls_pred.bias"):
            v = (rs.randn(*shape) * 0.5 + cls_bias).astype(np.float32)
        elif key.endswith("reg_pred.bias"):
            v = (rs.randn(*shape) * 0.5 + 1.0).astype(np.float32)
        else:  # conv / pred weights
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.2 if "cls_pred." in key else 0.25 if "reg_pred." in key else 0.4 * g_conv if ".dwconv." in key else g_conv
            v = (rs.randn(*shape) * (gain / math.sqrt(fan_in))).astype(np.float32)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v))
    return sd


def synth_images(batch, size, seed=1):
    """U[0,1) images [B,3,H,W] fp32, NumPy legacy RNG."""
    h, w = (size, size) if isinstance(size, int) else size
    return torch.from_numpy(np.random.RandomState(seed).rand(batch, 3, h, w).astype(np.float32))


# --------------------------------------------------------------------------------------------
# Train-form forward with eval-mode BatchNorm (branches kept separate)
# --------------------------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _conv_t(sd, p, x, k, stride=1):       # Conv.forward (common.py:46-47)
    return F.silu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, stride, k // 2)))


def _repvgg_t(sd, p, x):                  # RepVGGBlock.forward train branch (common.py:219-224)
    d = _bn(sd, p + ".rbr_dense.bn", F.conv2d(x, sd[p + ".rbr_dense.conv.weight"], None, 2, 1))
    o = _bn(sd, p + ".rbr_1x1.bn", F.conv2d(x, sd[p + ".rbr_1x1.conv.weight"], None, 2, 0))
    return F.relu(d + o)


def _unirep_t(sd, p, x, k):               # UniRepLKNetBlock.forward (common.py:3080-3083) + DilatedReparamBlock.forward (:3024-3031)
    c = x.shape[1]
    out = _bn(sd, p + ".dwconv.origin_bn", F.conv2d(x, sd[p + ".dwconv.lk_origin.weight"], None, 1, k // 2, 1, c))
    for kk in dil_branches(k):
        out = out + _bn(sd, p + ".dwconv.dil_bn_k%d_1" % kk,
                        F.conv2d(x, sd[p + ".dwconv.dil_conv_k%d_1.weight" % kk], None, 1, kk // 2, 1, c))
    return _bn(sd, p + ".norm", out)


def _graph_forward(scale, x, blocks):
    """Model.forward routing (yolo.py:186-201): y[i] kept for later `from` references."""
    y = []
    for i, f, op, a in arch(scale):
        if isinstance(f, list):
            inp = [y[i + j] if j < 0 else y[j] for j in f]
        elif i == 0:
            inp = x
        else:
            inp = y[i + f] if f < 0 else y[f]
        y.append(blocks(i, op, a, inp))
    return [y[31], y[32], y[33]]


def forward_train_form(sd, scale, x):
    """Heads' (stem, cls, reg) tuples from train-form weights, BN in eval mode."""
    def blocks(i, op, a, inp):
        p = "backbone.%d" % i
        if op == "repvgg":
            return _repvgg_t(sd, p, inp)
        if op == "rephdw":            # RepHDW.forward (common.py:938-946), DepthBottleneckUni.forward (:918-927)
            c_ = int(a["cout"] * 0.5)
            t = _conv_t(sd, p + ".conv1", inp, 1)
            outs = [t[:, :c_], t[:, c_:]]
            for d in range(a["depth"]):
                q = p + ".m.%d" % d
                z = _conv_t(sd, q + ".conv1", outs[-1], 1)
                z = F.silu(_unirep_t(sd, q + ".conv2", z, a["k"]))
                outs.append(_conv_t(sd, q + ".one_conv", z, 1))
            return _conv_t(sd, p + ".conv2", torch.cat(outs, 1), 1)
        if op == "mprep":             # MPRep.forward (common.py:786-792)
            x1 = _conv_t(sd, p + ".conv1", F.max_pool2d(inp, 2, 2), 1)
            return torch.cat([x1, _repvgg_t(sd, p + ".conv2", inp)], 1)
        if op == "sppf":              # SPPF.forward (common.py:122-129)
            t = _conv_t(sd, p + ".cv1", inp, 1)
            y1 = F.max_pool2d(t, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
            return _conv_t(sd, p + ".cv2", torch.cat([t, y1, y2, y3], 1), 1)
        if op == "cw":
            return _conv_t(sd, p + ".block", inp, 3, 2)
        if op == "concat":
            return torch.cat(inp, 1)
        if op == "up":
            return F.interpolate(inp, scale_factor=2, mode="nearest")
        if op == "head":              # Head_DepthUni.forward (common.py:1325-1336)
            t = _conv_t(sd, p + ".stem", inp, 1)
            cf = _conv_t(sd, p + ".cls_conv_s", _unirep_t(sd, p + ".cls_conv", t, a["k"]), 1)
            cls = torch.sigmoid(F.conv2d(cf, sd[p + ".cls_pred.weight"], sd[p + ".cls_pred.bias"]))
            rf = _conv_t(sd, p + ".reg_conv_s", _unirep_t(sd, p + ".reg_conv", t, a["k"]), 1)
            reg = F.conv2d(rf, sd[p + ".reg_pred.weight"], sd[p + ".reg_pred.bias"])
            return (t, cls, reg)
        raise ValueError(op)
    return _graph_forward(scale, x, blocks)


# --------------------------------------------------------------------------------------------
# Deploy switch: train-form state_dict -> {conv name: (weight, bias)}  
# --------------------------------------------------------------------------------------------
def _fuse(w, sd, bnp):
    """conv (no bias) + BN -> (w', b')   fuse_conv_and_bn (torch_utils.py:50-82) == fuse_bn (common.py:2636-2645)."""
    std = torch.sqrt(sd[bnp + ".running_var"] + BN_EPS)
    t = sd[bnp + ".weight"] / std
    return w * t.reshape(-1, 1, 1, 1), sd[bnp + ".bias"] - sd[bnp + ".running_mean"] * t


def _reparam_repvgg(sd, p):
    """get_equivalent_kernel_bias (common.py:226-230): 3x3 + zero-padded 1x1, biases summed."""
    k3, b3 = _fuse(sd[p + ".rbr_dense.conv.weight"], sd, p + ".rbr_dense.bn")
    k1, b1 = _fuse(sd[p + ".rbr_1x1.conv.weight"], sd, p + ".rbr_1x1.bn")
    return k3 + F.pad(k1, [1, 1, 1, 1]), b3 + b1


def _reparam_unirep(sd, p, k):
    """merge_dilated_branches (common.py:3033-3051) then fold the outer BN (common.py:3085-3100)."""
    w, b = _fuse(sd[p + ".dwconv.lk_origin.weight"], sd, p + ".dwconv.origin_bn")
    for kk in dil_branches(k):
        bw, bb = _fuse(sd[p + ".dwconv.dil_conv_k%d_1.weight" % kk], sd, p + ".dwconv.dil_bn_k%d_1" % kk)
        pad = k // 2 - kk // 2            # dilation 1: equivalent kernel == kernel (common.py:2940-2947)
        w = w + F.pad(bw, [pad] * 4)
        b = b + bb
    std = torch.sqrt(sd[p + ".norm.running_var"] + BN_EPS)
    s = sd[p + ".norm.weight"] / std
    return w * s.reshape(-1, 1, 1, 1), sd[p + ".norm.bias"] + (b - sd[p + ".norm.running_mean"]) * s


def reparam(sd, scale):
    """All deploy-form convs: name -> (weight fp32, bias fp32). 91 / 121 / 151 entries for n/s/m."""
    out = {}

    def conv(p):
        out[p + ".conv"] = _fuse(sd[p + ".conv.weight"], sd, p + ".bn")

    for i, f, op, a in arch(scale):
        p = "backbone.%d" % i
        if op == "repvgg":
            out[p + ".rbr_reparam"] = _reparam_repvgg(sd, p)
        elif op == "rephdw":
            conv(p + ".conv1")
            for d in range(a["depth"]):
                q = p + ".m.%d" % d
                conv(q + ".conv1")
                out[q + ".conv2.dwconv.lk_origin"] = _reparam_unirep(sd, q + ".conv2", a["k"])
                conv(q + ".one_conv")
            conv(p + ".conv2")
        elif op == "mprep":
            conv(p + ".conv1")
            out[p + ".conv2.rbr_reparam"] = _reparam_repvgg(sd, p + ".conv2")
        elif op == "sppf":
            conv(p + ".cv1"); conv(p + ".cv2")
        elif op == "cw":
            conv(p + ".block")
        elif op == "head":
            conv(p + ".stem")
            out[p + ".cls_conv.dwconv.lk_origin"] = _reparam_unirep(sd, p + ".cls_conv", a["k"])
            conv(p + ".cls_conv_s")
            out[p + ".reg_conv.dwconv.lk_origin"] = _reparam_unirep(sd, p + ".reg_conv", a["k"])
            conv(p + ".reg_conv_s")
            out[p + ".cls_pred"] = (sd[p + ".cls_pred.weight"], sd[p + ".cls_pred.bias"])
            out[p + ".reg_pred"] = (sd[p + ".reg_pred.weight"], sd[p + ".reg_pred.bias"])
    return out


# --------------------------------------------------------------------------------------------
# Deploy-form forward
# --------------------------------------------------------------------------------------------
def forward_deploy(dw, scale, x, taps=None):
    """Heads' (stem, cls, reg) tuples from deploy weights `dw` (= reparam(...)).

    `taps`, if a dict, receives every node output (for per-block parity tests)."""
    def c(name, x, stride=1, act="silu", groups=1):
        w, b = dw[name]
        y = F.conv2d(x, w, b, stride, w.shape[-1] // 2, 1, groups)
        return F.silu(y) if act == "silu" else F.relu(y) if act == "relu" else y

    def blocks(i, op, a, inp):
        p = "backbone.%d" % i
        if op == "repvgg":            # deploy forward (common.py:216-217): ReLU epilogue
            r = c(p + ".rbr_reparam", inp, 2, "relu")
        elif op == "rephdw":
            c_ = int(a["cout"] * 0.5)
            t = c(p + ".conv1.conv", inp)
            outs = [t[:, :c_], t[:, c_:]]
            for d in range(a["depth"]):
                q = p + ".m.%d" % d
                z = c(q + ".conv1.conv", outs[-1])
                z = c(q + ".conv2.dwconv.lk_origin", z, 1, "silu", z.shape[1])   # SiLU from DepthBottleneckUni.act
                outs.append(c(q + ".one_conv.conv", z))
            r = c(p + ".conv2.conv", torch.cat(outs, 1))
        elif op == "mprep":
            x1 = c(p + ".conv1.conv", F.max_pool2d(inp, 2, 2))
            r = torch.cat([x1, c(p + ".conv2.rbr_reparam", inp, 2, "relu")], 1)
        elif op == "sppf":
            t = c(p + ".cv1.conv", inp)
            y1 = F.max_pool2d(t, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
            r = c(p + ".cv2.conv", torch.cat([t, y1, y2, y3], 1))
        elif op == "cw":
            r = c(p + ".block.conv", inp, 2)
        elif op == "concat":
            r = torch.cat(inp, 1)
        elif op == "up":
            r = F.interpolate(inp, scale_factor=2, mode="nearest")
        elif op == "head":            # no activation between DW and 1x1 (common.py:1329,1333)
            t = c(p + ".stem.conv", inp)
            cf = c(p + ".cls_conv_s.conv", c(p + ".cls_conv.dwconv.lk_origin", t, 1, None, t.shape[1]))
            cls = torch.sigmoid(c(p + ".cls_pred", cf, 1, None))
            rf = c(p + ".reg_conv_s.conv", c(p + ".reg_conv.dwconv.lk_origin", t, 1, None, t.shape[1]))
            r = (t, cls, c(p + ".reg_pred", rf, 1, None))
        else:
            raise ValueError(op)
        if taps is not None:
            taps[i] = r
        return r
    return _graph_forward(scale, x, blocks)


# --------------------------------------------------------------------------------------------
# Detect head decode (eval branch)
# --------------------------------------------------------------------------------------------
def decode(heads, strides=STRIDES):
    """Detect_yaml.forward eval branch (yolo.py:355-396) -> [B, sum(L), 5+nc] fp32.

    anchors: (x+0.5, y+0.5) in grid units (anchor_generator.py:13-20); DFL expectation over 17 bins
    (yolo.py:377-378, proj = linspace(0,16,17) yolo.py:327-330); dist2bbox 'xywh' (general.py:29-40);
    box *= stride (yolo.py:389); objectness column = 1.0 (yolo.py:393)."""
    cls_l, box_l = [], []
    for (t, cls, reg), s in zip(heads, strides):
        b, _, h, w = t.shape
        l = h * w
        r = reg.reshape(b, 4, REG_MAX + 1, l).permute(0, 2, 1, 3)       # [B,17,4,L]
        prob = F.softmax(r, dim=1)
        proj = torch.linspace(0, REG_MAX, REG_MAX + 1, dtype=prob.dtype).reshape(1, -1, 1, 1)
        dist = (prob * proj).sum(1)                                      # [B,4,L]  ltrb
        dist = dist.permute(0, 2, 1)                                     # [B,L,4]
        sx = torch.arange(w, dtype=torch.float32) + 0.5
        sy = torch.arange(h, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        anc = torch.stack([xx, yy], -1).reshape(-1, 2)
        lt, rb = dist[..., :2], dist[..., 2:]
        x1y1 = anc - lt
        x2y2 = anc + rb
        box = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1) * float(s)
        box_l.append(box)
        cls_l.append(cls.reshape(b, cls.shape[1], l).permute(0, 2, 1))
    box = torch.cat(box_l, 1)
    cls = torch.cat(cls_l, 1)
    return torch.cat([box, torch.ones_like(box[..., :1]), cls], -1)


def predict(dw, scale, x):
    """Deploy-form Model.forward(x)[0] in eval mode."""
    return decode(forward_deploy(dw, scale, x))


# --------------------------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------------------------
def greedy_nms(boxes, scores, iou_threshold, float_threshold=False):
    """Restatement of torchvision.ops.nms' CPU kernel (called at nms.py:96) — PARITY UNPINNED.
    float_threshold=True: the comparison of torchvision's CUDA kernel instead (`float iou_threshold`: fp32 IoU > fl32(threshold)).

    boxes [n,4] xyxy fp32, scores [n] fp32. Stable descending sort of the scores (ties: lower index
    first), areas (x2-x1)*(y2-y1) in fp32, intersection with max(0,.) clamps, suppress j when
    inter/(area_i+area_j-inter) > iou_threshold with the fp32 quotient compared against the
    threshold in double.  Returns kept indices (int64) in descending-score order."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    s = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = b.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-s.astype(np.float64), kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = float(np.float32(iou_threshold)) if float_threshold else float(iou_threshold)
    for pos in range(n):
        i = order[pos]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def greedy_nms_torch(boxes, scores, iou_threshold):
    """torch-typed adaptor used as the `torchvision.ops.nms` stub when running the reference."""
    k = greedy_nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), iou_threshold)
    return torch.from_numpy(k).to(boxes.device)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, return_index=False, float_threshold=False):
    """NumPy restatement of yolov6/utils/nms.py:31-105 (time limit :101-103 dropped).

    prediction [B,N,5+nc] fp32.  Returns list of [n_i,6] fp32 arrays (x1,y1,x2,y2,conf,cls).
    With return_index also returns per image the flat candidate index box*nc+cls (multi_label) or
    box (best-class mode) of every survivor — the 'survivor indices' of the parity statement."""
    assert 0 <= conf_thres <= 1, f"conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided."
    assert 0 <= iou_thres <= 1, f"iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided."
    pred = np.asarray(prediction, dtype=np.float32)
    nc = pred.shape[2] - 5
    ct = np.float32(conf_thres)            # torch compares an fp32 tensor with a python scalar in fp32
    max_wh, max_nms = 4096, 30000
    multi_label = multi_label and nc > 1
    cand = np.logical_and(pred[..., 4] > ct, pred[..., 5:].max(-1) > ct)      # nms.py:48
    out, idx_out = [], []
    for bi in range(pred.shape[0]):
        sel = np.nonzero(cand[bi])[0]
        x = pred[bi][sel].copy()
        if x.shape[0] == 0:
            out.append(np.zeros((0, 6), np.float32)); idx_out.append(np.zeros((0,), np.int64)); continue
        x[:, 5:] *= x[:, 4:5]                                                   # nms.py:69
        box = np.empty((x.shape[0], 4), np.float32)                             # xywh2xyxy nms.py:21-28
        box[:, 0] = x[:, 0] - x[:, 2] / 2; box[:, 1] = x[:, 1] - x[:, 3] / 2
        box[:, 2] = x[:, 0] + x[:, 2] / 2; box[:, 3] = x[:, 1] + x[:, 3] / 2
        if multi_label:                                                         # nms.py:75-77 (row-major nonzero)
            bidx, cidx = np.nonzero(x[:, 5:] > ct)
            rows = np.concatenate([box[bidx], x[bidx, cidx + 5, None], cidx[:, None].astype(np.float32)], 1)
            flat = sel[bidx].astype(np.int64) * nc + cidx
        else:                                                                   # nms.py:78-80 (first max on ties)
            cidx = x[:, 5:].argmax(1)
            conf = x[np.arange(x.shape[0]), cidx + 5]
            rows = np.concatenate([box, conf[:, None], cidx[:, None].astype(np.float32)], 1)
            m = conf > ct
            rows, flat = rows[m], sel[m].astype(np.int64)
        if classes is not None:                                                 # nms.py:83-84
            m = np.isin(rows[:, 5], np.asarray(classes, dtype=np.float32))
            rows, flat = rows[m], flat[m]
        if rows.shape[0] == 0:
            out.append(np.zeros((0, 6), np.float32)); idx_out.append(np.zeros((0,), np.int64)); continue
        if rows.shape[0] > max_nms:                                             # nms.py:90-91 (ties: lower index first)
            o = np.argsort(-rows[:, 4].astype(np.float64), kind="stable")[:max_nms]
            rows, flat = rows[o], flat[o]
        off = rows[:, 5:6] * np.float32(0 if agnostic else max_wh)              # nms.py:94
        keep = greedy_nms(rows[:, :4] + off, rows[:, 4], iou_thres, float_threshold)[:max_det]   # nms.py:95-98
        out.append(rows[keep]); idx_out.append(flat[keep])
    return (out, idx_out) if return_index else out


# ---------------------------------------------------------------------------------------------------------------------
# Post-NMS tail: Evaler.scale_coords / box_convert / convert_to_coco_format, yolov6/core/evaler.py:374-434
# ---------------------------------------------------------------------------------------------------------------------
def scale_coords(coords, shape0, ratio_pad, scale_exact=False):
    """evaler.py:382-409 with ratio_pad given (the only branch the reference can execute: its `ratio_pad is None` branch
    multiplies an int by a list).  coords [n,4] fp32 xyxy in network-input pixels -> original-image pixels, clamped.
    Scalars enter the fp32 tensor arithmetic as fp32, exactly as torch does."""
    c = np.array(coords, dtype=np.float32, copy=True)
    gain, pad = ratio_pad
    gx = np.float32(gain[1] if scale_exact else gain[0])
    gy = np.float32(gain[0])
    c[:, [0, 2]] = (c[:, [0, 2]] - np.float32(pad[0])) / gx
    c[:, [1, 3]] = (c[:, [1, 3]] - np.float32(pad[1])) / gy
    c[:, [0, 2]] = np.clip(c[:, [0, 2]], np.float32(0), np.float32(shape0[1]))
    c[:, [1, 3]] = np.clip(c[:, [1, 3]], np.float32(0), np.float32(shape0[0]))
    return c


def coco_rows(outputs, shapes, image_ids, ids, scale_exact=False):
    """evaler.py:411-434: list of per-image [n,6] (x1,y1,x2,y2,conf,cls) -> (image_id [R], category_id [R], bbox [R,4] =
    x,y,w,h rounded to 3 decimals, score [R] rounded to 5), images with no detection skipped."""
    iid, cid, bb, sc = [], [], [], []
    for i, pred in enumerate(outputs):
        pred = np.asarray(pred, dtype=np.float32)
        if pred.shape[0] == 0:
            continue
        xy = scale_coords(pred[:, :4], shapes[i][0], shapes[i][1], scale_exact)
        cx = (xy[:, 0] + xy[:, 2]) / np.float32(2); cy = (xy[:, 1] + xy[:, 3]) / np.float32(2)     # box_convert :374-381
        w = xy[:, 2] - xy[:, 0]; h = xy[:, 3] - xy[:, 1]
        x = cx - w / np.float32(2); y = cy - h / np.float32(2)                                      # :420
        box = np.stack([x, y, w, h], 1).astype(np.float64)
        bb.append(np.round(box * 1000.0) / 1000.0)               # == Python round(v, 3) for doubles that come from fp32 (v*1000 is exact)
        sc.append(np.round(pred[:, 4].astype(np.float64) * 100000.0) / 100000.0)
        cid.append(np.asarray(ids)[pred[:, 5].astype(np.int64)])
        iid.append(np.full(pred.shape[0], image_ids[i], dtype=np.int64))
    if not bb:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 4)), np.zeros(0)
    return np.concatenate(iid), np.concatenate(cid).astype(np.int64), np.concatenate(bb), np.concatenate(sc)


# ---------------------------------------------------------------------------------------------------------------------
# Training loss: ComputeLoss with the task-aligned assigner — yolov6/models/loss.py:56-193 (formal-assigner
# branch, epoch >= warmup_epoch), yolov6/assigners/tal_assigner.py:21-151, assigner_utils.py:25-89, figure_iou.py (GIoU),
# yolov6/utils/general.py:29-49 (dist2bbox / bbox2dist), anchor_generator.py:26-53.  Written per image and per ground-truth box
# (the reference builds dense [B, n_max, A] masks); padded boxes of the reference never win an argmax nor a top-k, so they vanish.
# ---------------------------------------------------------------------------------------------------------------------
def train_anchors(feat_hw, strides=(8, 16, 32), offset=0.5):
    """anchor_generator.py:26-53: anchor centres in pixels and the stride of every anchor, levels concatenated."""
    pts, st = [], []
    for (h, w), s in zip(feat_hw, strides):
        ys = (torch.arange(h, dtype=torch.float32) + offset) * s
        xs = (torch.arange(w, dtype=torch.float32) + offset) * s
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pts.append(torch.stack([xx, yy], -1).reshape(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def _pair_iou(g, p, eps=1e-9):
    """assigner_utils.py:68-89: IoU of one gt box [4] with predicted boxes [A,4]."""
    x1y1 = torch.maximum(g[:2], p[:, :2]); x2y2 = torch.minimum(g[2:], p[:, 2:])
    inter = (x2y2 - x1y1).clip(0).prod(-1)
    a1 = (g[2:] - g[:2]).clip(0).prod(-1); a2 = (p[:, 2:] - p[:, :2]).clip(0).prod(-1)
    return inter / (a1 + a2 - inter + eps)


def tal_assign(pd_scores, pd_bboxes, anc_points, gts, nc=80, topk=13, alpha=1.0, beta=6.0, eps=1e-9):
    """One image.  pd_scores [A,nc] (sigmoid outputs), pd_bboxes [A,4] xyxy pixels, gts [n,5] = (label, x1,y1,x2,y2) pixels.
    -> target_labels [A] (label of gt 0 for background anchors, as the reference leaves it), target_bboxes [A,4], target_scores [A,nc], fg [A]."""
    A = pd_scores.shape[0]
    n = gts.shape[0]
    if n == 0:
        return torch.zeros(A, dtype=torch.long), torch.zeros(A, 4), torch.zeros(A, nc), torch.zeros(A, dtype=torch.bool)
    labels = gts[:, 0].long()
    ov = torch.stack([_pair_iou(gts[g, 1:], pd_bboxes, eps) for g in range(n)])                    # [n,A]
    metric = pd_scores[:, labels].t().pow(alpha) * ov.pow(beta)                                    # tal_assigner.py:96-111
    d = torch.cat([anc_points[None] - gts[:, None, 1:3], gts[:, None, 3:5] - anc_points[None]], -1)
    in_gts = (d.min(-1)[0] > eps).float()                                                          # assigner_utils.py:25-44
    mask_pos = torch.zeros(n, A)
    for g in range(n):                                                                             # tal_assigner.py:113-128
        idx = torch.topk(metric[g] * in_gts[g], topk, largest=True)[1]
        mask_pos[g, idx] = 1.0
    mask_pos = mask_pos * in_gts
    fg = mask_pos.sum(0)
    multi = fg > 1                                                                                 # assigner_utils.py:46-66
    if bool(multi.any()):
        best = ov.argmax(0)
        mask_pos[:, multi] = F.one_hot(best[multi], n).t().float()
        fg = mask_pos.sum(0)
    gt_idx = mask_pos.argmax(0)
    t_labels = labels[gt_idx]
    t_boxes = gts[gt_idx, 1:]
    t_scores = F.one_hot(t_labels, nc).float() * (fg > 0).float()[:, None]
    am = metric * mask_pos                                                                         # tal_assigner.py:66-71
    pos_am = am.max(-1, keepdim=True)[0]
    pos_ov = (ov * mask_pos).max(-1, keepdim=True)[0]
    norm = (am * pos_ov / (pos_am + eps)).max(0)[0]
    return t_labels, t_boxes, t_scores * norm[:, None], fg > 0


def atss_assign(anchors, n_level, pd_bboxes, gts, nc=80, topk=9):
    """ATSSAssigner.forward for one image (yolov6/assigners/atss_assigner.py:17-161), the warm-up assigner of ComputeLoss (loss.py:83-91).
    anchors [A,4] = the 5-stride anchor boxes of generate_anchors (anchor_generator.py:27-51), n_level = anchors per level, pd_bboxes [A,4]
    xyxy pixels, gts [n,5] = (label, x1,y1,x2,y2) pixels.  Per box: the `topk` anchors of every level nearest to its centre are candidates;
    positives are the candidates whose IoU (anchor box vs box) exceeds mean + std of the candidates' IoUs and whose centre lies inside
    the box; an anchor claimed by several boxes goes to the one with the largest anchor IoU; the score target is the IoU of the
    PREDICTED box with the assigned box.  -> target_labels, target_bboxes, target_scores, fg."""
    A = anchors.shape[0]
    n = gts.shape[0]
    if n == 0:
        return torch.zeros(A, dtype=torch.long), torch.zeros(A, 4), torch.zeros(A, nc), torch.zeros(A, dtype=torch.bool)
    labels = gts[:, 0].long()
    gb = gts[:, 1:]
    valid = (gb.sum(-1, keepdim=True) > 0).float()                                                 # loss.py:77 mask_gt
    # iou2d_calculator.bbox_overlaps (iou2d_calculator.py:63-246), mode 'iou', eps 1e-6
    area1 = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1]); area2 = (anchors[:, 2] - anchors[:, 0]) * (anchors[:, 3] - anchors[:, 1])
    wh = (torch.min(gb[:, None, 2:], anchors[None, :, 2:]) - torch.max(gb[:, None, :2], anchors[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    ov = inter / torch.max(area1[:, None] + area2[None] - inter, torch.tensor(1e-6))               # [n,A]
    gc = torch.stack([(gb[:, 0] + gb[:, 2]) / 2.0, (gb[:, 1] + gb[:, 3]) / 2.0], 1)                # assigner_utils.py:4-23
    ac = torch.stack([(anchors[:, 0] + anchors[:, 2]) / 2.0, (anchors[:, 1] + anchors[:, 3]) / 2.0], 1)
    dist = (gc[:, None] - ac[None]).pow(2).sum(-1).sqrt()
    in_cand = torch.zeros(n, A)
    cand_idx = []
    start = 0
    for nl in n_level:                                                                             # atss_assigner.py:89-116
        k = min(topk, nl)
        idx = dist[:, start:start + nl].topk(k, dim=-1, largest=False)[1] + start
        cand_idx.append(idx)
        in_cand.scatter_(1, idx, 1.0)
        start += nl
    cand_idx = torch.cat(cand_idx, 1)
    cand_ov = ov.gather(1, cand_idx)                                                               # atss_assigner.py:118-137
    thr = cand_ov.mean(-1, keepdim=True) + cand_ov.std(-1, keepdim=True)
    is_pos = torch.where(torch.where(in_cand > 0, ov, torch.zeros_like(ov)) > thr, in_cand, torch.zeros_like(in_cand))
    d = torch.cat([ac[None] - gb[:, None, :2], gb[:, None, 2:] - ac[None]], -1)
    in_gts = (d.min(-1)[0] > 1e-9).float()
    mask_pos = is_pos * in_gts * valid
    fg = mask_pos.sum(0)
    multi = fg > 1                                                                                 # assigner_utils.py:46-66
    if bool(multi.any()):
        best = ov.argmax(0)
        mask_pos[:, multi] = F.one_hot(best[multi], n).t().float()
        fg = mask_pos.sum(0)
    gt_idx = mask_pos.argmax(0)
    t_labels = labels[gt_idx]
    t_boxes = gb[gt_idx]
    t_scores = F.one_hot(t_labels, nc).float() * (fg > 0).float()[:, None]
    iou_pd = torch.stack([_pair_iou(gb[g], pd_bboxes) for g in range(n)]) * mask_pos              # atss_assigner.py:80-84
    return t_labels, t_boxes, t_scores * iou_pd.max(0)[0][:, None], fg > 0


def train_anchor_boxes(feat_hw, strides=(8, 16, 32), cell_size=5.0, offset=0.5):
    """anchor_generator.py:27-38: the square anchor boxes (side cell_size * stride) ATSS measures IoU and centre distance with."""
    pts, st = train_anchors(feat_hw, strides, offset)
    half = cell_size * st * 0.5
    return torch.cat([pts - half, pts + half], -1)


def _giou_loss(b1, b2, eps=1e-10):
    """figure_iou.py IOUloss(box_format='xyxy', iou_type='giou', eps=1e-10): rows of b1/b2 [M,4] -> [M,1]."""
    x1, y1, x2, y2 = b1.split(1, -1); u1, v1, u2, v2 = b2.split(1, -1)
    inter = (torch.min(x2, u2) - torch.max(x1, u1)).clamp(0) * (torch.min(y2, v2) - torch.max(y1, v1)).clamp(0)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = u2 - u1, v2 - v1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(x2, u2) - torch.min(x1, u1); ch = torch.max(y2, v2) - torch.min(y1, v1)
    c_area = cw * ch + eps
    return 1.0 - (iou - (c_area - union) / c_area)


def compute_loss(feat_hw, pred_scores, pred_distri, targets, nc=80, img_size=640, reg_max=16, strides=(8, 16, 32),
                 weights=(1.0, 2.5, 0.5), return_assignment=False, assigner="tal"):
    """loss.py:56-193 with the task-aligned assigner (epochs >= warmup_epoch) or, assigner="atss", the warm-up assigner (loss.py:83-91).  pred_scores [B,A,nc] in (0,1), pred_distri [B,A,4*(reg_max+1)] logits, targets
    [T,6] = (image, class, cx, cy, w, h) normalised to the image.  -> (loss, [iou, dfl, cls] weighted items)."""
    B, A, _ = pred_scores.shape
    pts, st = train_anchors(feat_hw, strides)
    pts_s = pts / st
    proj = torch.linspace(0, reg_max, reg_max + 1)
    dist = F.softmax(pred_distri.view(B, A, 4, reg_max + 1), -1).matmul(proj)                       # loss.py:190-193
    pred_boxes = torch.cat([pts_s - dist[..., :2], pts_s + dist[..., 2:]], -1)                      # dist2bbox, xyxy in stride units
    t_labels = torch.zeros(B, A, dtype=torch.long); t_boxes = torch.zeros(B, A, 4)
    t_scores = torch.zeros(B, A, nc); fg = torch.zeros(B, A, dtype=torch.bool)
    for b in range(B):
        rows = targets[targets[:, 0] == b]
        g = torch.zeros(rows.shape[0], 5)
        if rows.shape[0]:
            xywh = rows[:, 2:6] * img_size                                                          # loss.py:179-188
            g[:, 0] = rows[:, 1]
            g[:, 1] = xywh[:, 0] - xywh[:, 2] / 2; g[:, 2] = xywh[:, 1] - xywh[:, 3] / 2
            g[:, 3] = xywh[:, 0] + xywh[:, 2] / 2; g[:, 4] = xywh[:, 1] + xywh[:, 3] / 2
        if assigner == "atss":
            t_labels[b], t_boxes[b], t_scores[b], fg[b] = atss_assign(train_anchor_boxes(feat_hw, strides), [h * w for h, w in feat_hw],
                                                                      (pred_boxes[b] * st).detach(), g, nc)
        else:
            t_labels[b], t_boxes[b], t_scores[b], fg[b] = tal_assign(pred_scores[b].detach(), (pred_boxes[b] * st).detach(), pts, g, nc)
    t_boxes = t_boxes / st                                                                          # loss.py:152
    lab = torch.where(fg, t_labels, torch.full_like(t_labels, nc))
    one_hot = F.one_hot(lab, nc + 1)[..., :-1].float()
    w = 0.75 * pred_scores.pow(2.0) * (1 - one_hot) + t_scores * one_hot                            # VarifocalLoss, loss.py:200-206
    loss_cls = (F.binary_cross_entropy(pred_scores.float(), t_scores.float(), reduction="none") * w).sum()
    tss = t_scores.sum()
    loss_cls = loss_cls / tss
    if bool(fg.any()):                                                                              # BboxLoss, loss.py:217-267
        bw = t_scores.sum(-1)[fg].unsqueeze(-1)
        loss_iou = (_giou_loss(pred_boxes[fg], t_boxes[fg]) * bw).sum() / tss
        ltrb = torch.cat([pts_s - t_boxes[..., :2], t_boxes[..., 2:] - pts_s], -1).clip(0, reg_max - 0.01)[fg]   # bbox2dist
        pd = pred_distri.view(B, A, 4, reg_max + 1)[fg]
        tl = ltrb.long(); tr = tl + 1
        wl = tr.float() - ltrb; wr = 1 - wl
        ce = lambda t: F.cross_entropy(pd.reshape(-1, reg_max + 1), t.reshape(-1), reduction="none").view(tl.shape)
        loss_dfl = (((ce(tl) * wl + ce(tr) * wr).mean(-1, keepdim=True)) * bw).sum() / tss
    else:
        loss_iou = torch.tensor(0.); loss_dfl = torch.tensor(0.)
    loss = weights[0] * loss_cls + weights[1] * loss_iou + weights[2] * loss_dfl
    items = torch.stack([weights[1] * loss_iou, weights[2] * loss_dfl, weights[0] * loss_cls]).detach()
    if return_assignment:
        return loss, items, (t_labels, t_boxes, t_scores, fg)
    return loss, items
