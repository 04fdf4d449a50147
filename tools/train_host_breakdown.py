"""Host time inside the bodies of the train_ops autograd Functions (forward and backward), per class, for one step — the backward bodies run on
the autograd engine's thread where cProfile of the main thread does not see them:  python tools/train_host_breakdown.py [scale] [batch]"""
import collections
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
T = importlib.import_module("maf-yolo_amd.train_ops")
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(cls, name):
    f = getattr(cls, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            r = acc["%s.%s" % (cls.__name__, name)]
            r[0] += time.perf_counter() - t0
            r[1] += 1
    setattr(cls, name, staticmethod(g))


for cls in (T._Conv1x1, T._Conv3x3s2, T._Conv1x1s2, T._DWConv, T._BNAct, T._MaxPool):
    wrap(cls, "forward")
    wrap(cls, "backward")
dev = torch.device("cuda:0")
model = M.Model(scale)
model.load_state_dict(synth.synth_state_dict(model, scale, 0))
model = model.to(dev).train()
opt = M.build_optimizer(model, lr0=0.005)
scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
x = synth.synth_images(B, 640, seed=1).to(dev)
g_ = torch.Generator().manual_seed(1)
wh = torch.rand(7 * B, 2, generator=g_) * 0.35 + 0.04
ctr = wh / 2 + torch.rand(7 * B, 2, generator=g_) * (1 - wh)
targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g_).float(), ctr, wh], 1).to(dev)
crit = M.ComputeLoss(warmup_epoch=0)


def step():
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    loss = crit((feats, cls, reg), targets, 0, 0)[0]
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for _ in range(4):
    step()
torch.cuda.synchronize()
acc.clear()
N = 5
tot = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
print("issue time %.2f ms/step" % (tot / N * 1e3))
s = 0.0
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %6.2f ms/step  %4d calls/step  %6.1f us/call" % (k, t / N * 1e3, n // N, t / n * 1e6))
    s += t
print("inside the Function bodies: %.2f ms/step" % (s / N * 1e3))
