"""How deep into the score-sorted candidate list does the greedy scan go before max_det survivors exist (bench workload)?
python tools/nms_depth.py"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
import bench as B
dev = torch.device("cuda:0")
m = M.Model("n"); m.load_state_dict(synth.synth_state_dict(m, "n", 0)); m = m.to(dev).eval().half()
x = synth.synth_images(32, 640, seed=1).to(dev).half()
B.calibrate_cls_bias(m, x, 2000, M, torch)
with torch.no_grad():
    pred = m(x)[0]
dets = M.non_max_suppression(pred, 0.03, 0.65, multi_label=True)
sc = (pred[..., 5:] * pred[..., 4:5]).float()
res = []
for b in range(32):
    s = sc[b].flatten(); cand = s[s > 0.03]
    n = cand.numel(); kept = dets[b].shape[0]
    last = dets[b][-1, 4].item() if kept else 1.0
    depth = int((cand >= last).sum())
    cls = (sc[b] > 0.03).sum(0)
    res.append((n, kept, depth, int(cls.max())))
print("n  kept  depth(of the last survivor)  largest class")
for r in res[:12]: print(r)
a = np.array(res); print("mean", a.mean(0), "max", a.max(0))
