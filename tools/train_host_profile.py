"""cProfile of the host side of the train step (where do the ~30 ms of Python per step go?): python tools/train_host_profile.py [scale] [batch]"""
import cProfile
import importlib
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
model = M.Model(scale)
model.load_state_dict(synth.synth_state_dict(model, scale, 0))
model = model.to(dev).train()
opt = M.build_optimizer(model, lr0=0.01 / 64 * B)
scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
x = synth.synth_images(B, 640, seed=1).to(dev)
g = torch.Generator().manual_seed(100)
wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
crit = M.ComputeLoss(warmup_epoch=0)


def step():
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    loss = crit((feats, cls, reg), targets, 0, 0)[0]
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))
