import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from maf_yolo_amd import lib
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
for (H, cin, cout) in [(160, 48, 48), (160, 24, 72), (80, 64, 192), (80, 192, 576), (80, 576, 192), (40, 384, 1152), (40, 576, 384), (20, 768, 768), (80, 256, 256), (40, 192, 576), (80, 128, 384)]:
    M = int(os.environ.get('BS', '16')) * H * H
    x = torch.randn(M, cin, device='cuda').half(); dy = torch.randn(M, cout, device='cuda').half()
    reps = int(os.environ.get("REPS", "1"))
    dw = torch.zeros(reps, cout, cin, device='cuda')
    args = (x.data_ptr(), cin, dy.data_ptr(), cout, M, cin, cout, lib.F16, dw.data_ptr()) + ((reps,) if reps > 1 or os.environ.get("REPS") else ()) + (st,)
    for _ in range(3): lib.check(L.maf_conv1x1_wgrad(*args))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): lib.check(L.maf_conv1x1_wgrad(*args))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    t1 = time.perf_counter()
    for _ in range(10): ref = dy.t() @ x
    torch.cuda.synchronize(); dtt = (time.perf_counter() - t1) / 10
    print('wgrad M=%d %d->%d: %.1f us (%.2f TB/s)   torch.mm %.1f us' % (M, cin, cout, dt * 1e6, (x.numel() + dy.numel()) * 2 / dt / 1e12, dtt * 1e6))
