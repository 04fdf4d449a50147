import sys, ctypes as C; sys.path.insert(0,'.')
import torch, maf_yolo_amd as M
from maf_yolo_amd import synth, lib
sys.argv=['x']
import bench
dev=torch.device('cuda:0')
model=M.Model('n'); model.load_state_dict(synth.synth_state_dict(model,'n',0)); model=model.to(dev).eval()
x=synth.synth_images(32,640,1).to(dev).half()
bench.calibrate_cls_bias(model,x,2000,M,torch)
with torch.no_grad(): pred=model(x)[0]
for i in range(3):
    M.non_max_suppression(pred,0.03,0.65,multi_label=True)
torch.cuda.synchronize()
buf=(C.c_uint64*8)(); lib.check(lib.load().maf_nms_debug(buf)); print('cycles sort,A,B,total,n,kept,wide,t_ld', list(buf))
