"""Experiment: one batch of 32 as S concurrent micro-batches of 32/S on S streams.  python tools/microbatch_exp.py"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = importlib.import_module("maf-yolo_amd")
synth = importlib.import_module("maf-yolo_amd.synth")
dev = torch.device("cuda:0")
m = M.Model("n"); m.load_state_dict(synth.synth_state_dict(m, "n", 0)); m = m.to(dev).eval().half(); m.autotune = True
x = synth.synth_images(32, 640, seed=1).to(dev).half()
for S in (1, 2, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    parts = [p.contiguous() for p in x.chunk(S)]
    ev0 = torch.cuda.Event(); 
    def run(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                cur = torch.cuda.current_stream(dev)
                e = torch.cuda.Event(); e.record(cur)
                outs = []
                for k in range(S):
                    streams[k].wait_event(e)
                    with torch.cuda.stream(streams[k]):
                        outs.append(m(parts[k], slot=k)[0])
                for k in range(S):
                    cur.wait_stream(streams[k])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(5)
    print("S=%d micro-batches of %d: %.4f ms per 32 images" % (S, 32 // S, run(40)), flush=True)
