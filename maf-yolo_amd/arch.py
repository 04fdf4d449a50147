"""Graph description of MAF-YOLO-{n,s,m} and of any model given in the reference's YAML schema.

The reference builds its graph with `parse_model` (yolov6/models/yolo.py:15-120) from
configs/yaml/MAF-YOLO-*.yaml: rows `[from, number, module, args]`, channel widths scaled by
`width_multiple` for some module types and literal for others.  `nodes_from_yaml_dict` applies the
same rules for the module types MAF-YOLO uses, so a user's reference YAML loads unchanged;
`builtin(scale)` produces the three released architectures without any file (the reference tree
does not exist on the GPU box).
"""
import math
from dataclasses import dataclass, field
from typing import List, Union


@dataclass
class Node:
    i: int
    f: Union[int, List[int]]       # 'from' exactly as in the YAML (-1 = previous node)
    kind: str                      # repvgg | rephdw | mprep | sppf | cw | concat | up | head | out
    cin: Union[int, List[int]] = 0
    cout: int = 0
    args: dict = field(default_factory=dict)

    def sources(self):
        """Absolute indices of the producer nodes."""
        fs = self.f if isinstance(self.f, list) else [self.f]
        return [self.i + x if x < 0 else x for x in fs]


def make_divisible(x, divisor):
    return int(math.ceil(x / divisor) * divisor)      # yolo.py:220-222


_KIND = {"RepVGGBlock": "repvgg", "RepHDW": "rephdw", "MPRep": "mprep", "SPPF": "sppf", "ConvWrapper": "cw",
         "Concat": "concat", "nn.Upsample": "up", "Head_DepthUni": "head", "Out": "out"}


def nodes_from_yaml_dict(d, ch_in=3, nc=80):
    """[from, number, module, args] rows -> Node list, with parse_model's width/depth rules."""
    gd, gw = d["depth_multiple"], d["width_multiple"]
    rows = list(d["backbone"]) + list(d.get("neck", [])) + list(d["effidehead"])
    ch, nodes = [], []
    for i, (f, n, m, args) in enumerate(rows):
        m = m if isinstance(m, str) else getattr(m, "__name__", str(m))
        if m not in _KIND:
            raise NotImplementedError("module type %r is not part of the MAF-YOLO hot path (SURVEY.md §8a)" % m)
        kind = _KIND[m]
        n = max(round(n * gd), 1) if n > 1 else n                       # yolo.py:27
        fs = f if isinstance(f, list) else [f]
        cins = [ch_in if (i == 0) else ch[i + x if x < 0 else x] for x in fs]
        a = {}
        if kind == "repvgg":
            cout = make_divisible(args[0] * gw, 4)                      # yolo.py:28-30
            if args[1:] != [3, 2]:
                raise NotImplementedError("RepVGGBlock other than 3x3 stride 2 (SURVEY.md §0 fact 1)")
        elif kind == "sppf":
            cout = make_divisible(args[0] * gw, 4)
            a = dict(k=args[1] if len(args) > 1 else 5)
            if a["k"] != 5:
                raise NotImplementedError("SPPF kernel other than 5")
        elif kind == "rephdw":                                          # yolo.py:36-40: literal channels, n -> depth
            cout = args[0]
            a = dict(depth=n, expansion=args[2], k=args[3], depth_expansion=args[4])
        elif kind == "mprep":
            cout = make_divisible(args[0] * gw, 8)                      # yolo.py:92-96
        elif kind == "cw":
            cout = args[0]                                              # yolo.py:59-62: literal
            if list(args[1:]) != [3, 2]:
                raise NotImplementedError("ConvWrapper other than 3x3 stride 2")
        elif kind == "head":
            cout = make_divisible(args[0] * gw, 8)                      # yolo.py:56-59
            a = dict(reg_max=args[1], k=args[2], nc=nc)
        elif kind == "concat":
            cout = sum(cins)
        elif kind == "up":
            cout = cins[0]
            if list(args[1:]) != [2, "nearest"]:
                raise NotImplementedError("Upsample other than nearest x2")
        else:  # out
            cout = cins[0]
        nodes.append(Node(i=i, f=f, kind=kind, cin=cins if isinstance(f, list) else cins[0], cout=cout, args=a))
        ch.append(cout)
    return nodes


# (width_multiple, per-row literals) of the three released models — configs/yaml/MAF-YOLO-{n,s,m}.yaml
_BUILTIN = {
    "n": dict(gw=0.375, hdw=[(48, 1), (96, 1), (192, 1), (384, 1)], cw=(96, 64, 64, 128, 128),
              neck=[(192, 1), (128, 1), (128, 1), (128, 1), (128, 1), (192, 1)], head=(341, 341, 512)),
    "s": dict(gw=0.5, hdw=[(64, 2), (128, 2), (256, 2), (512, 2)], cw=(128, 96, 96, 192, 192),
              neck=[(256, 2), (192, 2), (192, 2), (192, 2), (192, 2), (256, 2)], head=(384, 384, 512)),
    "m": dict(gw=0.75, hdw=[(96, 2), (192, 4), (384, 4), (768, 2)], cw=(256, 192, 192, 192, 192),
              neck=[(512, 3), (384, 3), (384, 3), (256, 3), (384, 3), (384, 3)], head=(341, 512, 512)),
}


def builtin_yaml_dict(scale):
    """The released architecture as a dict in the reference's YAML schema (generated, not read from a file)."""
    t = _BUILTIN[scale]
    hdw, cw, nk = t["hdw"], t["cw"], t["neck"]

    def H(c_d, k, f=-1, sc=False):
        return [f, c_d[1], "RepHDW", [c_d[0], sc, 0.5, k, 3]]

    backbone = [[-1, 1, "RepVGGBlock", [64, 3, 2]], [-1, 1, "RepVGGBlock", [128, 3, 2]], H(hdw[0], 3, sc=True),
                [-1, 1, "MPRep", [256]], H(hdw[1], 5, sc=True), [-1, 1, "MPRep", [512]], H(hdw[2], 7, sc=True),
                [-1, 1, "MPRep", [1024]], H(hdw[3], 9, sc=True), [-1, 1, "SPPF", [1024, 5]]]
    up = [-1, 1, "nn.Upsample", [None, 2, "nearest"]]
    neck = [[6, 1, "ConvWrapper", [cw[0], 3, 2]], [[-1, 9], 1, "Concat", [1]], H(nk[0], 9), list(up),
            [4, 1, "ConvWrapper", [cw[1], 3, 2]], [[-1, 6, -2], 1, "Concat", [1]], H(nk[1], 7), list(up),
            [2, 1, "ConvWrapper", [cw[2], 3, 2]], [[-1, 4, -2], 1, "Concat", [1]], H(nk[2], 5),
            [[-1, 17], 1, "Concat", [1]], H(nk[3], 5),
            [-1, 1, "ConvWrapper", [cw[3], 3, 2]], [20, 1, "ConvWrapper", [cw[3], 3, 2]],
            [[-2, -1, 16, 13], 1, "Concat", [1]], H(nk[4], 7),
            [-1, 1, "ConvWrapper", [cw[4], 3, 2]], [16, 1, "ConvWrapper", [cw[4], 3, 2]],
            [[-2, -1, 12], 1, "Concat", [1]], H(nk[5], 9)]
    head = [[22, 1, "Head_DepthUni", [t["head"][0], 16, 5]], [26, 1, "Head_DepthUni", [t["head"][1], 16, 7]],
            [30, 1, "Head_DepthUni", [t["head"][2], 16, 9]], [[31, 32, 33], 1, "Out", []]]
    return dict(depth_multiple=1, width_multiple=t["gw"], backbone=backbone, neck=neck, effidehead=head)


def builtin(scale, nc=80):
    return nodes_from_yaml_dict(builtin_yaml_dict(scale), 3, nc)


def dil_branch_kernels(k):
    """Parallel small depth-wise kernels of a DilatedReparamBlock, all dilation 1 (common.py:2997-3008)."""
    table = {9: (7, 5, 3), 7: (5, 3), 5: (3, 1), 3: (3, 1)}
    if k not in table:
        raise NotImplementedError("DilatedReparamBlock kernel size %d is not used by MAF-YOLO" % k)
    return table[k]
