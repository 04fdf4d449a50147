import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
B = 32
dev = torch.device("cuda:0")
model = M.Model("n"); model.load_state_dict(synth.synth_state_dict(model, "n", 0)); model = model.to(dev).train()
opt = M.build_optimizer(model, lr0=0.01)
scaler = torch.amp.GradScaler("cuda")
x = synth.synth_images(B, 640, seed=1).to(dev)
g = torch.Generator().manual_seed(1)
wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
crit = M.ComputeLoss(warmup_epoch=0)
T = {}
def step():
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    t1 = time.perf_counter()
    loss = crit((feats, cls, reg), targets, 0, 0)[0]
    t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    t3 = time.perf_counter()
    scaler.step(opt); scaler.update()
    t4 = time.perf_counter()
    for k, v in (("fwd", t1 - t0), ("loss", t2 - t1), ("bwd", t3 - t2), ("opt", t4 - t3)):
        T[k] = T.get(k, 0) + v
for _ in range(3): step()
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(10): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("back to back: host %.1f ms/step, total %.1f ms/step" % (th * 100, tt * 100), {k: round(v * 100, 2) for k, v in T.items()})
# issue time of a step against an EMPTY queue (the host never waits for the GPU): synchronise before every step
T.clear()
iss = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    iss += time.perf_counter() - t0
torch.cuda.synchronize()
print("issue only : host %.1f ms/step" % (iss * 100), {k: round(v * 100, 2) for k, v in T.items()})
