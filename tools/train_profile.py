"""torch.profiler table of one training step (which aten ops the non-native time goes to): python tools/train_profile.py [scale] [batch]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
scale = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.to(dev).train()
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
    scaler.scale(loss).backward(); scaler.step(opt); scaler.update()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50, max_shapes_column_width=90))
