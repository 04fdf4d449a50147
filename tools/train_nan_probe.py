import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
scale, B = sys.argv[1], int(sys.argv[2]); fused = sys.argv[3] == "1"
dev = torch.device("cuda:0")
model = M.Model(scale); model.load_state_dict(synth.synth_state_dict(model, scale, 0)); model = model.to(dev).train()
opt = M.build_optimizer(model, lr0=0.01 / 64 * B, fused=fused)
scaler = torch.amp.GradScaler("cuda")
x = synth.synth_images(B, 640, seed=1).to(dev)
g = torch.Generator().manual_seed(100)
wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
crit = M.ComputeLoss(warmup_epoch=0)
for i in range(int(os.environ.get('STEPS', '12'))):
    with torch.autocast("cuda", dtype=torch.float16):
        (feats, cls, reg), _ = model(x)
    loss = crit((feats, cls, reg), targets, 0, 0)[0]
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    gn = sum(float(p.grad.float().abs().sum()) for p in model.parameters() if p.grad is not None)
    scaler.step(opt); scaler.update()
    bad = sum(int(not torch.isfinite(p).all()) for p in model.parameters())
    print(i, float(loss), "cls finite", bool(torch.isfinite(cls).all()), "reg finite", bool(torch.isfinite(reg).all()), "grad sum", gn, "scale", scaler.get_scale(), "bad params", bad)
