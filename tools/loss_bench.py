"""Time the pieces of the device-side ComputeLoss (f2) on synthetic head outputs: python tools/loss_bench.py [boxes_per_image]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("maf-yolo_amd")
loss_mod = importlib.import_module("maf-yolo_amd.loss")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    dev = torch.device("cuda:0")
    B, A, nc = 32, 8400, 80
    g = torch.Generator().manual_seed(0)
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for dt in ((torch.float16,) if only else (torch.float16, torch.float32)):
        scores = torch.sigmoid(torch.randn(B, A, nc, generator=g) * 1.5 - 3).to(dev, dt).requires_grad_(True)
        distri = (torch.randn(B, A, 68, generator=g)).to(dev, dt).requires_grad_(True)
        feats = [torch.zeros(B, 8, s, s, device=dev) for s in (80, 40, 20)]
        n = per * B
        wh = torch.rand(n, 2, generator=g) * 0.35 + 0.04
        ctr = wh / 2 + torch.rand(n, 2, generator=g) * (1 - wh)
        targets = torch.cat([torch.arange(B).repeat_interleave(per)[:, None].float(), torch.randint(0, nc, (n, 1), generator=g).float(), ctr, wh], 1).to(dev)
        for fused in ([True] if only else [False, True]):
            crit = M.ComputeLoss(ori_img_size=640, warmup_epoch=0, fused=fused)
            def fwd():
                return crit((feats, scores, distri), targets, 0, 0)[0]
            def fwdbwd():
                scores.grad = None; distri.grad = None
                fwd().backward()
            pts, st = loss_mod._anchors(feats, (8, 16, 32), 0.5, dev)
            boxes = torch.rand(B, A, 4, device=dev) * 100
            boxes[..., 2:] += boxes[..., :2]
            sc32 = scores.detach().float()
            def assign():
                loss_mod.task_aligned_assign(sc32, boxes, pts, targets, B, 640, nc)
            if only:
                print("fwd+bwd %.0f us" % timeit(fwdbwd, 50)); continue
            print("%s fused=%s boxes/img=%d: assign(+gathers) %.0f us, loss fwd %.0f us, fwd+bwd %.0f us" % (dt, fused, per, timeit(assign), timeit(fwd), timeit(fwdbwd)), flush=True)

main()
