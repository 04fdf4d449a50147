"""Import harness for the read-only reference at /root/reference (build container only).

Never used on the GPU box: nothing under tests/ (-m gpu), bench.py or smoke() imports this.
It follows the verified recipe of SURVEY.md Appendix C: four tiny stub modules are injected
so that `yolov6.models.yolo`, `yolov6.utils.torch_utils` and `yolov6.utils.nms` import with
torchvision / cv2 / timm / addict absent.  The greedy-NMS inner kernel that the reference gets
from torchvision (yolov6/utils/nms.py:96) is supplied by `nms_fn` (our own restatement): that
boundary is therefore "parity unpinned"  and says so wherever it is used.
"""
import os
import sys
import types
import runpy
import contextlib

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def available():
    return os.path.isdir(os.path.join(REF, "yolov6"))


def load(nms_fn):
    """Returns a namespace with the reference's Model / fuse_model / blocks / non_max_suppression."""
    import torch.nn as nn

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    if "yolov6.models.yolo" not in sys.modules:
        _stub("timm"); _stub("timm.models"); _stub("timm.models.layers", DropPath=DropPath)
        _stub("cv2", setNumThreads=lambda n: None)
        tv = _stub("torchvision", __version__="0")
        tv.ops = _stub("torchvision.ops", nms=nms_fn)
        _stub("addict", Dict=AttrDict)
        sys.path.insert(0, REF)
    else:
        sys.modules["torchvision.ops"].nms = nms_fn
    import logging
    logging.disable(logging.CRITICAL)
    with _cwd(REF):
        from yolov6.models.yolo import Model
        from yolov6.utils.torch_utils import fuse_model
        from yolov6.layers import common
        from yolov6.utils.nms import non_max_suppression
    ns = types.SimpleNamespace(Model=Model, fuse_model=fuse_model, common=common,
                               non_max_suppression=non_max_suppression)
    return ns


def build(ns, scale):
    """Train-form reference model for scale in {'n','s','m'} (eval mode)."""
    with _cwd(REF):
        cfg = AttrDict({k: v for k, v in runpy.run_path("configs/MAF-YOLO-%s.py" % scale).items()
                        if not k.startswith("__")})
        with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
            model = ns.Model(cfg, channels=3, num_classes=80, anchors=cfg.model.head.anchors).eval()
    return model


def to_deploy(ns, model):
    """Mirrors checkpoint.py:83-93 + evaler.py:101-109 (fuse_model, switch_to_deploy, reparameterize)."""
    import copy
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        d = ns.fuse_model(copy.deepcopy(model)).eval()
        for l in d.modules():
            if isinstance(l, ns.common.RepVGGBlock):
                l.switch_to_deploy()
            if isinstance(l, ns.common.UniRepLKNetBlock):
                l.reparameterize()
    return d
