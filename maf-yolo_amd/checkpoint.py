"""Checkpoint bridge: the reference's `.pt` files <-> this package's Model.

The reference checkpoints are pickled *modules*, not state_dicts: `torch.save({'model': deepcopy(model).half(), 'ema': ...,
'updates', 'optimizer', 'epoch'})` (yolov6/core/engine.py:195-201, stripped by checkpoint.py:107-122) and
`load_checkpoint` (checkpoint.py:83-93) does `ckpt['ema' if ckpt.get('ema') else 'model'].float()` + `fuse_model`.
Unpickling them normally needs the reference's source tree on sys.path.  Here a restricted unpickler maps every class
that is not on a short explicit allow-list (tensor rebuild functions, storages, dtypes, plain containers) — the `yolov6.*` layers, the
torch.nn module classes, the config Dict, any builtins / functools / torch.hub callable — to an inert stand-in, so the module tree
arrives as plain objects whose `_parameters` / `_buffers` / `_modules` dictionaries are walked into a state_dict; the
architecture comes from the pickled model's own `yaml` dict (yolo.py:144-146).  No reference code is imported or executed.

    model = load_checkpoint("MAFYOLOn.pt")            # maf_yolo_amd.Model, eval mode, fp32 masters (same call as the reference's)
    state = reference_state_dict(model)               # what the reference's Model.load_state_dict(strict=True) accepts
"""
import collections
import io
import pickle

import torch

from .model import Model

# Explicit allow-list: ONLY these globals are resolved to the real objects; everything else a pickle names — the
# reference's `yolov6.*` layers, torch.nn module classes, builtins.eval / exec / getattr / __import__, functools.partial,
# torch.hub / torch.utils loaders, os / subprocess ... — becomes an inert stand-in that stores state and runs nothing.
_ALLOWED = {
    "collections": {"OrderedDict"},
    "builtins": {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"},
    "__builtin__": {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "object"},
    "_codecs": {"encode"},                                     # protocol-2 encoding of bytes objects
    "copyreg": {"_reconstructor"},                             # object.__new__(cls) for classes resolved through THIS lookup
    "torch._utils": {"_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_parameter_with_state"},
    "torch.nn.parameter": {"Parameter"},
    "torch": {"Size", "device", "Tensor"},                     # + storage classes and dtype objects, see _allowed()
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"},
    "numpy._core.multiarray": {"_reconstruct", "scalar"},
}


def _allowed(module, name):
    if name in _ALLOWED.get(module, ()):
        return True
    if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
        return True                                            # torch.FloatStorage ..., torch.float16 ...
    return False


class _Inert:
    """Stand-in for a class of the reference's source tree: keeps whatever state pickle hands it, runs nothing."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):     # (dict_state, slots_state)
            self.__dict__.update(state[0] or {})
            self.__dict__.update(state[1])

    def __call__(self, *a, **k):
        return self


class _InertDict(dict):
    """Stand-in for dict subclasses (addict.Dict configs): pickle fills it through dict's own protocol."""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)


_made = {}


def _standin(module, name):
    key = (module, name)
    cls = _made.get(key)
    if cls is None:
        base = _InertDict if name in ("Dict", "Config", "ConfigDict", "AttrDict") else _Inert
        cls = _made[key] = type(name, (base,), {"__module__": "maf_yolo_amd.checkpoint", "_ref_class": module + "." + name})
    return cls


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if _allowed(module, name):
            return super().find_class(module, name)
        return _standin(module, name)


class _PickleModule:
    """`pickle_module` for torch.load: stock pickle with the restricted class lookup."""
    __name__ = "pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _Unpickler(io.BytesIO(b), **kw).load())
    dumps, dump, PickleError, UnpicklingError = pickle.dumps, pickle.dump, pickle.PickleError, pickle.UnpicklingError


def _walk(obj, prefix, out):
    d = getattr(obj, "__dict__", {})
    skip = d.get("_non_persistent_buffers_set", ())
    for k, v in (d.get("_parameters") or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if torch.is_tensor(v) else v
    for k, v in (d.get("_buffers") or {}).items():
        if v is not None and k not in skip:
            out[prefix + k] = v
    for k, m in (d.get("_modules") or {}).items():
        if m is not None:
            _walk(m, prefix + k + ".", out)


def read_reference_checkpoint(weights, map_location="cpu"):
    """-> (state_dict with the reference's key names, yaml dict of the architecture, number of classes, raw ckpt dict)."""
    ckpt = torch.load(weights, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    root = ckpt
    if isinstance(ckpt, dict):
        root = ckpt["ema"] if ckpt.get("ema") is not None else ckpt["model"]          # checkpoint.py:87
    sd = collections.OrderedDict()
    if isinstance(root, dict):                                                        # a plain state_dict was saved
        sd.update(root)
        return sd, None, None, ckpt
    _walk(root, "", sd)
    yaml_dict = getattr(root, "yaml", None)
    det = (getattr(root, "__dict__", {}).get("_modules") or {}).get("detect")
    nc = getattr(det, "nc", None) if det is not None else None
    return sd, yaml_dict, nc, ckpt


def load_checkpoint(weights, map_location=None, inplace=True, fuse=True):
    """Counterpart of yolov6/utils/checkpoint.py:83-93: the checkpoint's EMA (else model) weights in a Model, fp32, eval mode.
    `fuse` is accepted for signature compatibility: the deploy algebra runs when the HIP plan is built (layers.py:fused)."""
    sd, yaml_dict, nc, _ = read_reference_checkpoint(weights, map_location or "cpu")
    if yaml_dict is None:                                                             # a bare state_dict: the released scales only
        n = len(sd)
        scale = {838: "n", 1206: "s", 1568: "m"}.get(n)
        if scale is None:
            raise ValueError("checkpoint holds a bare state_dict with %d tensors: not MAF-YOLO n/s/m" % n)
        config, nc = scale, 80
    else:
        config = {k: yaml_dict[k] for k in ("depth_multiple", "width_multiple", "backbone", "neck", "effidehead") if k in yaml_dict}
    model = Model(config, channels=3, num_classes=nc or 80)
    missing = model.load_state_dict({k: v.float() if torch.is_floating_point(v) else v for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.float().eval()


def reference_state_dict(model):
    """Trained weights under the reference's parameter names (what its Model.load_state_dict(strict=True) takes)."""
    return collections.OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())
