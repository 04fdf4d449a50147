import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maf_yolo_amd as M
from maf_yolo_amd import synth, train_ops
dev = torch.device("cuda:0")
def run(freeze):
    model = M.Model("n"); model.load_state_dict(synth.synth_state_dict(model, "n", 0)); model = model.to(dev).train()
    if freeze:
        for n, p in model.named_parameters():
            if p.dim() == 4: p.requires_grad_(False)          # conv weights: no weight-gradient kernels
    ex = M.GradExchange(model)
    opt = M.build_optimizer(model, lr0=0.005, momentum=0.937, weight_decay=5e-4)
    scaler = torch.amp.GradScaler("cuda")
    B = 32
    x = synth.synth_images(B, 640, seed=1).to(dev)
    g = torch.Generator().manual_seed(100)
    wh = torch.rand(7 * B, 2, generator=g) * 0.35 + 0.04
    ctr = wh / 2 + torch.rand(7 * B, 2, generator=g) * (1 - wh)
    targets = torch.cat([torch.arange(B).repeat_interleave(7)[:, None].float(), torch.randint(0, 80, (7 * B, 1), generator=g).float(), ctr, wh], 1).to(dev)
    crit = M.ComputeLoss(ori_img_size=640, warmup_epoch=0)
    def step():
        with torch.autocast("cuda", dtype=torch.float16):
            (feats, cls, reg), _ = model(x)
        loss = crit((feats, cls, reg), targets, 0, 0)[0]
        ex.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    tp = [e[1] for e in model._tapes.values()]
    print("freeze conv weights", freeze, "ms/step %.3f" % ms, "tape", tp and tp[0].ready, tp and tp[0].failed)
    ex.close()
run(False); run(True)
