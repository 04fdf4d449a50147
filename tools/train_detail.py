"""Per-launch HIP-event times of the native kernels of ONE training step, with shapes / strides: python tools/train_detail.py [scale] [batch] [kind-substring]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
train_ops = importlib.import_module("maf-yolo_amd.train_ops")
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
want = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
model = M.Model(scale)
model.load_state_dict(synth.synth_state_dict(model, scale, 0))
model = model.to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
scaler = torch.amp.GradScaler("cuda")
x = synth.synth_images(B, 640, seed=1).to(dev)
g = torch.Generator().manual_seed(1)
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
train_ops.profile, train_ops.profile_detail = {}, []
step()
torch.cuda.synchronize()
train_ops.profile_collect()
rows = [r for r in train_ops.profile_detail if want in r[0]]
tot = sum(r[2] for r in rows)
print("%d launches, %.3f ms" % (len(rows), tot))
for kind, note, ms, nb in sorted(rows, key=lambda r: -r[2])[:int(os.environ.get("TOP", "60"))]:
    print("%-18s %-44s %8.1f us %8.1f MB %7.2f TB/s" % (kind, note, ms * 1e3, nb / 1e6, nb / ms / 1e9))
