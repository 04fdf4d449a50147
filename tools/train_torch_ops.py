import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
B = 32
dev = torch.device("cuda:0")
model = M.Model("n"); model.load_state_dict(synth.synth_state_dict(model, "n", 0)); model = model.to(dev).train()
opt = M.build_optimizer(model, lr0=0.005)
scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
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
for _ in range(4): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::upsample_nearest2d_backward", "aten::_to_copy", "aten::constant_pad_nd", "aten::index_put_", "aten::slice_backward"):
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:45]:
    print("%8.1f us  x%-3d %-28s %s" % r)
