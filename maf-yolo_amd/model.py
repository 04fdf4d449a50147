"""`Model` — the reference's model surface (yolov6/models/yolo.py:122-217) over the HIP engine.

    model = Model("n")                       # or Model(cfg) with the reference's config object / YAML dict
    model.load_state_dict(reference_state_dict)      # same 838 / 1206 / 1568 keys
    pred, featmaps = model.eval()(images)    # [B, A, 5+nc] fp32 (xywh px, obj = 1, class probs), as yolo.py:355-396

Eval-mode forward = one C call into libmafyolo_hip (engine.py): the re-parameterised graph, NHWC,
fp16 storage / fp32 accumulate when the input is fp16 (the `--half` path of evaler.py:112,162) or
fp32 storage when the input is fp32.  There is no CPU path: CPU tensors raise.  Training-mode
forward returns the reference's `(feats, cls[B,A,nc], reg[B,A,4*(reg_max+1)])` tuple from the
train-form modules in layers.py.
"""
import copy
import itertools
from operator import attrgetter

import torch
import torch.nn as nn

from .config import cfg
from . import arch, lib, train_ops
from .engine import Plan
from .layers import (RepVGGBlock, RepHDW, MPRep, SPPF, ConvWrapper, Concat, Head_DepthUni, Out)


class Upsample(nn.Upsample):
    """nn.Upsample whose output keeps the input's dtype under torch.autocast.  autocast runs upsample_nearest2d in fp32, the following
    torch.cat is promoted to fp32 and the conv behind it casts the whole concat back to fp16 (32 x 288 x 80 x 80: 155 us, forward and
    backward, four times per step): nearest-neighbour copies values, so staying in fp16 is bit-identical to that round trip."""

    def forward(self, x, out=None):
        if self.scale_factor == 2 and self.mode == "nearest" and x.is_cuda:
            return train_ops.upsample2x(x, out=out)                 # csrc/pool_train.hip: strided source, result straight into a concat buffer's slot
        with torch.autocast(device_type=x.device.type, enabled=False):
            return super().forward(x)


class Detect_yaml(nn.Module):
    """Holds the DFL projection and level strides (yolo.py:301-331). Decode itself is csrc/decode.hip."""

    def __init__(self, num_classes=80, anchors=1, num_layers=3, use_dfl=True, reg_max=16, stride=(8, 16, 32)):
        super().__init__()
        self.nc, self.no, self.nl = num_classes, num_classes + 5, num_layers
        self.na = anchors if not isinstance(anchors, (list, tuple)) else len(anchors[0]) // 2
        self.use_dfl, self.reg_max = use_dfl, reg_max
        self.register_buffer("stride", torch.tensor(list(stride), dtype=torch.float32), persistent=False)
        self.proj_conv = nn.Conv2d(reg_max + 1, 1, 1, bias=False)
        self.initialize_biases()

    def initialize_biases(self):               # yolo.py:327-330
        self.proj = nn.Parameter(torch.linspace(0, self.reg_max, self.reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view(1, self.reg_max + 1, 1, 1).clone().detach(), requires_grad=False)

    def forward(self, heads, val_loss=False, logits=False):
        """Training branch (yolo.py:333-354): flatten + permute + cat of the per-level (stem, cls, reg).  logits=True: h[1] are the class LOGITS and the head's sigmoid
        (common.py:1332) is applied here — on HIP tensors the whole branch is one launch per direction (train_ops.detect_join, csrc/detect_join.hip)."""
        if logits:
            cls, reg = train_ops.detect_join(heads)
        else:
            cls = torch.cat([h[1].flatten(2).permute(0, 2, 1) for h in heads], 1)
            reg = torch.cat([h[2].flatten(2).permute(0, 2, 1) for h in heads], 1)
        return [h[0] for h in heads], cls, reg

    @staticmethod
    def level_views(feats, cls, reg):
        """[(stem, cls_l [B,nc,h,w], reg_l [B,4*(reg_max+1),h,w])] per level as VIEWS of the joined tensors: the `featmaps` the reference returns beside them
        (yolo.py:205-209; its callers discard them: engine.py:150, evaler.py:168)."""
        out, a0 = [], 0
        for f in feats:
            h, w = f.shape[-2:]
            out.append((f, cls[:, a0:a0 + h * w].permute(0, 2, 1).unflatten(2, (h, w)), reg[:, a0:a0 + h * w].permute(0, 2, 1).unflatten(2, (h, w))))
            a0 += h * w
        return out


def _nodes_from_config(config, num_classes):
    if isinstance(config, str) and config in ("n", "s", "m"):
        return arch.builtin(config, num_classes), dict(strides=(8, 16, 32), reg_max=16, use_dfl=True)
    if isinstance(config, dict) and "backbone" in config:
        return arch.nodes_from_yaml_dict(copy.deepcopy(config), 3, num_classes), dict(strides=(8, 16, 32), reg_max=16, use_dfl=True)
    # the reference's Config object: config.model.yaml_file + config.model.head.{strides,use_dfl,reg_max}
    model_cfg = getattr(config, "model", None)
    if model_cfg is not None and getattr(model_cfg, "yaml_file", None):
        import yaml
        with open(model_cfg.yaml_file, encoding="ascii", errors="ignore") as f:
            d = yaml.safe_load(f)
        head = model_cfg.head
        return (arch.nodes_from_yaml_dict(d, d.get("ch", 3), num_classes),
                dict(strides=tuple(head.strides), reg_max=getattr(head, "reg_max", 16), use_dfl=getattr(head, "use_dfl", True)))
    raise ValueError("Model(config): pass 'n' | 's' | 'm', a YAML dict in the reference's schema, or the reference's config object")


_VERSION = attrgetter("_version")


class _Aliases(list):
    """The aliases train_ops.fanout made of a node's output: every reader pops one."""


class Model(nn.Module):
    def __init__(self, config="n", channels=3, num_classes=80, anchors=1, precision=None, dispatch="engine"):
        super().__init__()
        self._maf_apply_gen = 0                # counts `_apply` calls (.to() / .cuda() / .half() ...): solver.ModelEMA's cheap staleness check
        assert channels == 3, "MAF-YOLO takes 3-channel images"
        self.nodes, hcfg = _nodes_from_config(config, num_classes)
        assert hcfg["use_dfl"] and hcfg["reg_max"] == 16 or hcfg["use_dfl"], "DFL head expected (configs/MAF-YOLO-n.py:15-16)"
        layers = []
        for nd in self.nodes:
            a = nd.args
            if nd.kind == "repvgg":
                m = RepVGGBlock(nd.cin, nd.cout)
            elif nd.kind == "rephdw":
                m = RepHDW(nd.cin, nd.cout, a["depth"], a["expansion"], a["k"], a["depth_expansion"])
            elif nd.kind == "mprep":
                m = MPRep(nd.cin, nd.cout)
            elif nd.kind == "sppf":
                m = SPPF(nd.cin, nd.cout, a["k"])
            elif nd.kind == "cw":
                m = ConvWrapper(nd.cin, nd.cout, 3, 2)
            elif nd.kind == "concat":
                m = Concat(1)
            elif nd.kind == "up":
                m = Upsample(None, 2, "nearest")
            elif nd.kind == "head":
                m = Head_DepthUni(nd.cin, nd.cout, a["reg_max"], a["k"], a["nc"])
            elif nd.kind == "out":
                m = Out()
            else:
                raise NotImplementedError(nd.kind)
            m.i, m.f, m.type = nd.i, nd.f, type(m).__name__
            layers.append(m)
        self.backbone = nn.Sequential(*layers)
        self.save = sorted({s for nd in self.nodes for s in nd.sources() if nd.i > 0 and s != nd.i - 1} |
                           {s for nd in self.nodes if isinstance(nd.f, list) for s in nd.sources()})
        self.detect = Detect_yaml(num_classes, anchors, 3, hcfg["use_dfl"], hcfg["reg_max"], hcfg["strides"])
        self.nc = num_classes
        self.names = [str(i) for i in range(num_classes)]
        self.build_type = "yaml"
        self.precision = precision            # None: follow the input dtype; "fp16" | "fp32" force it
        self._plans = {}
        self._plans_version = None            # weight fingerprint the cached plans were packed from (see weights_version)
        self._fp_tensors = None
        self.fuse_bottlenecks = "auto"        # fused DepthBottleneckUni kernel (csrc/bottleneck.hip): True / False / "auto" (measured per layer when autotune is on)
        self.fuse_stem = True                 # True / 2: backbone.0 + backbone.1 + the 1x1 that opens backbone.2 in one launch (csrc/stem2.hip; fp16 plans of n and s); 1: without the 1x1; False
        self.fuse_head = "auto"               # per level {cls,reg}_conv_s -> pred -> sigmoid / DFL decode in one launch (csrc/head_tail.hip; fp16, 80 classes)
        self.fuse_tail = "auto"               # the 1x1 conv that closes a RepHDW block inside the launch of its last fully fused bottleneck (csrc/bottleneck.hip, op.nc): "auto" = blocks of one bottleneck (every block of n) / True = wherever the instantiation exists (+ s / m's two-bottleneck blocks: opt-in, engine.Plan) / False
        self.fuse_mprep = "auto"              # MPRep (MaxPool2d + 1x1 | 3x3 s2) in one launch (csrc/conv3s2_lds.hip, 48 / 64 channels, big maps): True / False / "auto" = when autotune is on
        self.twin_convs = True                # the two equal side convs of a MAFPN level (backbone.23 / .24, .27 / .28) as one launch
        self.autotune = False                 # True: time the MFMA tile candidates of every conv when an fp16 plan is built
        self.multi_stream = False             # False | 1 (heads) | 2 (heads + neck side convs): independent branches on separate HIP streams inside the engine
        self.step_tape = "auto" if cfg.step_tape else False      # training: replay the recorded launch lists of a step (tape.py) once a batch shape has been seen RECORD_AT times under a GradExchange; False: always eager
        assert dispatch in ("engine", "ops")
        self.dispatch = dispatch              # "engine": one C call per forward (engine.py) | "ops": the graph op by op through torch.ops.mafyolo (ops_forward.py)

    # ------------------------------------------------------------------ nn.Module plumbing
    @property
    def stride(self):
        return self.detect.stride

    def train(self, mode=True):
        if mode:
            self._plans = {}                  # weights are about to change
        elif getattr(self, "training", False):
            self.release_tapes()              # train -> eval (the evaluation between epochs, engine.py:173-222): the step tapes pin a whole step's activations per batch shape
        return super().train(mode)

    def release_tapes(self):
        """Drop the recorded step tapes (tape.py) and what they pin: every activation and scratch of a full step per batch shape (several GB for m), the padded
        gradient buffers registered in train_ops.zero_padded.  The next RECORD_AT train-mode forwards of a shape run eagerly and record again.  Outputs of a replayed
        forward are views of the tape's static buffers (valid until the next forward of that shape): callers that keep predictions across steps clone them."""
        for ent in getattr(self, "_tapes", {}).values():
            tp = ent[1]
            if tp is not None:
                tp.release()
        self._tapes = {}

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        """Drop every cached plan (they hold packed copies of the weights).  Called automatically when the weight fingerprint
        changes; call it by hand after replacing a parameter's storage (`p.data = ...`), which leaves no trace in `_version`."""
        self._plans = {}
        self._plans_version = None
        self._fp_tensors = None
        self._pack_plan = None                 # training: the staged weight transforms point at the old parameter storage
        self.release_tapes()                   # ... and so do the recorded launch lists
        self._ops_weights = (None, None)

    def _apply(self, fn, *a, **k):
        # .to() / .cuda() / .float() / .half()-style conversions replace buffers and parameter storage (yolo.py:211-215 moves
        # detect.stride the same way): plans packed from the old tensors are stale
        self.invalidate()
        self._maf_apply_gen = getattr(self, "_maf_apply_gen", 0) + 1             # solver.ModelEMA: its cached tensor lists are stale from here on
        return super()._apply(fn, *a, **k)

    def weights_version(self):
        """Fingerprint of the parameters and buffers: the sum of their autograd version counters, which every in-place update bumps —
        an optimizer step, `ema.update` (the reference evaluates `self.ema.ema`, mutated in place every step and never put in train
        mode: engine.py:246), `load_state_dict`, manual `p.mul_()` edits.  ~60 us for the 839 tensors of n."""
        ts = self._fp_tensors
        if ts is None:
            ts = self._fp_tensors = list(itertools.chain(self.parameters(), self.buffers()))
        return sum(map(_VERSION, ts)) + len(ts)

    def half(self):
        """Reference callers do `model.half()` after the deploy switch (evaler.py:112). Masters stay fp32 here;
        fp16 plans are packed from them (fuse in fp32, then cast — the reference's order)."""
        self.precision = "fp16"
        return self

    def float(self):
        self.precision = None
        return super().float()

    def __deepcopy__(self, memo):
        plans, self._plans = self._plans, {}
        fpt, self._fp_tensors = self._fp_tensors, None
        pp, self._pack_plan = getattr(self, "_pack_plan", None), None
        tp, self._tapes = getattr(self, "_tapes", {}), {}
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            self._plans, self._fp_tensors, self._pack_plan, self._tapes = plans, fpt, pp, tp
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_plans"] = {}
        d["_fp_tensors"] = d["_plans_version"] = d["_pack_plan"] = None
        d["_tapes"] = {}
        return d

    # ------------------------------------------------------------------ forward
    def _forward_train_form(self, x, raw_heads=False, logits=False):
        if getattr(self, "_pack_plan", None) is None:
            self._pack_plan = train_ops.PackPlan()
        train_ops.begin_step(self._pack_plan, x.device)          # every weight transform of the step in one launch (train_ops.PackPlan)
        # Concat nodes without a copy (train_ops.CatBuffer): an input whose producer can store anywhere (the last BatchNorm apply pass of a ConvWrapper /
        # RepHDW / SPPF, the up-sampling kernel) goes straight into its slot of the concat's buffer — of the FIRST concat that lists it; every other input
        # (a map a second concat lists) is copied into its slot by `join`.
        res = getattr(self, "_cat_residents", None)
        if res is None:
            res = self._cat_residents = {}
            for nd, m in zip(self.nodes, self.backbone):
                if isinstance(m, Concat) and m.d == 1:
                    srcs = nd.sources()
                    widths = [self.nodes[s].cout for s in srcs]
                    for slot, s in enumerate(srcs):
                        if s not in res and isinstance(self.backbone[s], (ConvWrapper, RepHDW, SPPF, Upsample)) and srcs.count(s) == 1:
                            res[s] = (nd.i, slot, widths)
        bufs = {}

        def slot_of(j, slot, widths):
            def f(z):
                cb = bufs.get(j)
                if cb is None:
                    cb = bufs[j] = train_ops.CatBuffer(z, widths)
                return cb.slot(slot)
            return f

        # a map several later nodes read (every backbone level feeds the neck two to four times): one alias per reader, their gradients summed by ONE launch
        # (train_ops.fanout) instead of the autograd engine's add kernel per extra reader
        uses = getattr(self, "_node_uses", None)
        if uses is None:
            uses = self._node_uses = {}
            for nd in self.nodes:
                if nd.i > 0:
                    for j in nd.sources():
                        uses[j] = uses.get(j, 0) + 1
        y = []
        n_head, n_cw, on_lane = 0, 0, {}
        for nd, m in zip(self.nodes, self.backbone):
            if nd.i > 0:
                for j in nd.sources():
                    if j in on_lane:                                              # its producer ran on a lane: the main stream takes the tensor over
                        y[j] = train_ops.lane_join(y[j], on_lane.pop(j))
                src = [y[j].pop() if isinstance(y[j], _Aliases) else y[j] for j in nd.sources()]
                x = src if isinstance(nd.f, list) else src[0]
            r = res.get(nd.i) if train_ops.cat_free else None
            if raw_heads and isinstance(m, ConvWrapper) and isinstance(x, torch.Tensor):
                # a recording step tape: the side convs of the neck (two equal, independent ones per level: backbone.23 / .24, .27 / .28) take turns on a lane and the
                # main stream; whoever reads the result first joins the lane (below)
                n_cw += 1
                out_ = None if r is None else slot_of(*r)
                x, ln = train_ops.lane_run(n_cw % 2, (lambda t, m=m, o=out_: m(t) if o is None else m(t, out=o)), x)
                if ln and uses.get(nd.i, 0) > 1:
                    x = train_ops.lane_join(x, ln)
                elif ln:
                    on_lane[nd.i] = ln
            elif r is not None:
                x = m(x, out=slot_of(*r))
            elif nd.i in bufs:
                x = train_ops.join(bufs.pop(nd.i), x)
            elif raw_heads and isinstance(m, Head_DepthUni):
                x = m(x, raw=True, lanes=(2 * n_head + 1, 2 * n_head + 2))      # a recording step tape: the six head branches on lanes of their own
                n_head += 1
            elif logits and isinstance(m, Head_DepthUni):
                x = m(x, raw=True)                                               # class logits: Detect's join applies the sigmoid (train_ops.detect_join)
            else:
                x = m(x)
            n_use = uses.get(nd.i, 0)
            if n_use > 1 and isinstance(x, torch.Tensor) and x.is_cuda and x.requires_grad and train_ops.cat_free:
                x = _Aliases(train_ops.fanout(x, n_use))
            y.append(x)
        if raw_heads:                         # (stem, logits, box, lane, lane): the main stream takes the two tensors over from their lanes
            x = [(h[0], train_ops.lane_join(h[1], h[3]), train_ops.lane_join(h[2], h[4])) for h in x]
        return x                              # list of three (stem, cls, reg)

    def _train_out(self, x):
        """(stem feature maps per level, cls [B,A,nc] probabilities, reg [B,A,4*(reg_max+1)]) of a train-mode forward (yolo.py:179-209 + 333-354): eager (layers.py op by
        op, then Detect's join), or — steady state of a training loop under a GradExchange — the recorded launch lists of tape.py."""
        tape = self._tape_for(x)
        if tape is None:
            native = x.is_cuda and not train_ops.framework_ops
            if x.is_cuda:
                x = x.contiguous(memory_format=torch.channels_last)      # NHWC in memory: what the HIP kernels take
            return tuple(self.detect(self._forward_train_form(x, logits=native), logits=native))
        if tape.ready:
            return tape.replay_forward(x)
        return tape.record_forward(self, x)

    def _tape_for(self, x):
        """The step tape this forward runs on (ready: replay; not ready: this call records), or None for the eager path."""
        from . import exchange, tape as _tape
        ex = exchange.current
        if (not getattr(self, "step_tape", "auto") or ex is None or not x.is_cuda or x.dim() != 4 or x.shape[1] != 3 or not torch.is_grad_enabled()
                or train_ops.profile is not None or train_ops.framework_ops or train_ops._deterministic or not train_ops.cat_free or train_ops._rec is not None):
            return None
        half = x.dtype == torch.float16 or (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16)
        if not half or x.dtype not in (torch.float16, torch.float32) or x.shape[2] % 32 or x.shape[3] % 32:
            return None
        if not hasattr(self, "_tapes"):
            self._tapes = {}
        key = (tuple(x.shape), x.dtype, x.device.index, id(ex), train_ops._stream(x.device), train_ops.wgrad_stream)
        ent = self._tapes.get(key)
        if ent is None:
            if len(self._tapes) >= 4:
                old = self._tapes.pop(next(iter(self._tapes)))
                if old[1] is not None:
                    old[1].release()
            ent = self._tapes[key] = [0, None, 0]                            # forwards seen, tape, recordings tried
        ent[0] += 1
        tp = ent[1]
        if tp is not None:
            if tp.ready:
                if tp.pending_backward:
                    if tp.graph_alive():                                         # a second forward before the first one's backward: the static buffers are taken
                        train_ops.stats["tape_busy_eager"] = train_ops.stats.get("tape_busy_eager", 0) + 1
                        return None
                    tp.drop_pending()                                            # that forward's graph is gone, no backward will come: the tape is free again
                return tp
            if tp.failed is not None or tp.phase is not None:
                return None                                                      # the recording was dropped (tp.failed says why): eager from here on
            ent[1] = tp = None                                                   # recorded forward whose backward never ran: try again
        pp = getattr(self, "_pack_plan", None)
        if (ent[0] < _tape.RECORD_AT or ent[2] >= 2 or pp is None or pp.dirty or pp.table is None                # (the staging plan of the weights must have settled:
                or any(not p_.is_cuda for p_ in self.parameters())):                                             #  its table upload is a torch copy)
            return None
        ent[2] += 1
        ent[1] = _tape.StepTape(ex, x.device)
        return ent[1]

    def plan_for(self, x, head_feats=False, slot=0):
        """head_feats: the caller wants the per-level cls / reg tensors (featmaps): plan without the fused head tail.
        slot: plans of different slots own different activation arenas, so forwards of different slots may be in flight at the same time
        on different HIP streams (a serving loop alternates slots: the small-map kernels of one batch run under the big-map kernels of
        the other)."""
        if not x.is_cuda:
            raise lib.MafError("MAF-YOLO eval forward runs on the HIP engine only: got a %s tensor (no CPU fallback)" % x.device)
        B, ch, H, W = x.shape
        assert ch == 3
        if x.dtype == torch.uint8:
            in_dt = lib.U8
        elif x.dtype == torch.float16:
            in_dt = lib.F16
        elif x.dtype == torch.float32:
            in_dt = lib.F32
        else:
            raise TypeError("image dtype must be uint8, float16 or float32")
        if self.precision == "fp16":
            dt = lib.F16
        elif self.precision == "fp32":
            dt = lib.F32
        else:
            dt = lib.F32 if x.dtype == torch.float32 else lib.F16
        fuse_head = bool(getattr(self, "fuse_head", True)) and not head_feats
        key = (B, H, W, dt, in_dt, x.device.index, fuse_head, slot, repr(getattr(self, "fuse_stem", True)), repr(self.fuse_bottlenecks), repr(getattr(self, "fuse_tail", "auto")), self.multi_stream, bool(getattr(self, "twin_convs", True)))
        ver = self.weights_version()
        if ver != self._plans_version:         # the weights changed in place since the cached plans were packed (EMA update, optimizer step ...)
            self._plans = {}
            self._plans_version = ver
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 8:
                self._plans.pop(next(iter(self._plans)))
            fuse = None
            if self.autotune and dt == lib.F16 and self.fuse_bottlenecks == "auto":
                from .engine import choose_fusion
                fuse = choose_fusion(self, B, H, W, dt, in_dt, x.device, x.contiguous())
            plan = Plan(self, B, H, W, dt, in_dt, x.device, fuse=fuse, fuse_head=fuse_head)
            if self.autotune and dt == lib.F16:
                plan.autotune(x.contiguous())
            self._plans[key] = plan
        return plan

    def traceable(self, dtype=torch.float16):
        """The eval forward as a pure function of the image — `f(x) -> pred [B, A, 5 + nc]`, x [B, 3, H, W] of `dtype` on the HIP device — made of
        torch.ops.mafyolo.* calls only (ops_forward.py), with the deploy-form weights of the CURRENT parameters captured: what
        `torch.compile(f, fullgraph=True)` / `torch.export` trace (Model.forward itself does weight-version bookkeeping in plain Python)."""
        from . import ops_forward, torch_ops
        weights = ops_forward.deploy_weights(self, dtype)
        ops = torch_ops.load()
        strides = [float(s) for s in self.detect.stride.tolist()]

        def f(x):
            return ops_forward.forward(self, x, weights, ops, strides)[0]
        return f

    def forward(self, x, val_loss=False, slot=0):
        if self.training:
            out = self._train_out(x)
            return [out, self.detect.level_views(*out)]
        if getattr(self, "dispatch", "engine") == "ops" and not val_loss:
            # the same graph as a sequence of torch.ops.mafyolo.* calls (ops_forward.py): what torch.compile / export can trace
            from . import ops_forward
            if not x.is_cuda:
                raise lib.MafError("MAF-YOLO eval forward runs on the HIP device only: got a %s tensor (no CPU fallback)" % x.device)
            dt = torch.float32 if (self.precision == "fp32" or (self.precision is None and x.dtype == torch.float32)) else torch.float16
            ver = (self.weights_version(), dt)
            if getattr(self, "_ops_weights", (None, None))[0] != ver:
                self._ops_weights = (ver, ops_forward.deploy_weights(self, dt))
            xin = (x.to(dt) / 255 if x.dtype == torch.uint8 else x.to(dt)).contiguous(memory_format=torch.channels_last)
            pred, heads = ops_forward.forward(self, xin, self._ops_weights[1])
            return [pred, [tuple(h) for h in heads]]
        plan = self.plan_for(x, head_feats=val_loss, slot=slot)
        x = x.contiguous()
        with torch.cuda.device(x.device):
            conf = getattr(self, "nms_filter", None)
            if conf is not None and not val_loss and 0.0 <= conf < 1.0 and plan.filter_ok():
                # the candidate filter of the NMS call that follows (same conf_thres, multi_label) runs inside the head tails, which hold the
                # class scores in registers anyway: nms.nms_raw finds the lists through the tensor and skips its pass over the prediction
                from . import nms as _nms
                pred = torch.empty(plan.B, plan.A, 5 + plan.nc, dtype=torch.float32, device=plan.device)
                ws = _nms.candidate_workspace(plan.device, plan.B, plan.A, plan.nc, slot)
                plan.run_into(x, pred, cand=(ws, conf))
                pred._maf_cand = (ws, float(conf), ws._maf_gen, pred._version)   # generation + version counter: nms.nms_raw checks both
            else:
                pred = plan.run(x, graph=False)      # hipGraph replay needs fixed buffers: use Plan.run_into(x, pred, graph=True)
        feats = plan.featmaps()
        if val_loss:
            return [self.detect(feats), feats]
        return [pred, feats]
