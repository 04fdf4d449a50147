"""Whole train step (forward, loss, backward, fused SGD under the GradScaler) captured in ONE hipGraph and replayed: does it capture, does it
match eager, what does a replayed step cost?   python tools/graph_step_try.py [scale] [batch]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")


def make():
    torch.manual_seed(0)
    model = M.Model(scale)
    model.load_state_dict(synth.synth_state_dict(model, scale, 0))
    model = model.to(dev).train()
    opt = M.build_optimizer(model, lr0=0.01 / 64 * B)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    return model, opt, scaler


x = synth.synth_images(B, 640, seed=1).to(dev)
g = torch.Generator().manual_seed(100)
wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
crit = M.ComputeLoss(warmup_epoch=0)


def step(model, opt, scaler):
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    loss = crit((feats, cls, reg), targets, 0, 0)[0]
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return loss.detach()


# eager reference trajectory
model, opt, scaler = make()
eager = [float(step(model, opt, scaler)) for _ in range(8)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(model, opt, scaler)
torch.cuda.synchronize()
print("eager  %.2f ms/step" % ((time.perf_counter() - t0) * 100), ["%.4f" % v for v in eager])

model, opt, scaler = make()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
losses = []
with torch.cuda.stream(s):
    for _ in range(3):
        losses.append(float(step(model, opt, scaler)))
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    static_loss = step(model, opt, scaler)
torch.cuda.synchronize()
for _ in range(5):
    graph.replay()
    losses.append(float(static_loss))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    graph.replay()
torch.cuda.synchronize()
print("graph  %.2f ms/step" % ((time.perf_counter() - t0) * 100), ["%.4f" % v for v in losses])
