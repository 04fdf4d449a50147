import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from maf_yolo_amd import lib
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
for (H, C, k) in [(160, 72, 3), (80, 144, 5), (80, 192, 3), (40, 288, 7), (20, 576, 9), (80, 128, 5)]:
    x = torch.randn(32, H, H, C, device='cuda').half(); dy = torch.randn(32, H, H, C, device='cuda').half()
    dw = torch.zeros(32, C, k * k, device='cuda')
    for _ in range(3): lib.check(L.maf_dw_wgrad(x.data_ptr(), C, dy.data_ptr(), C, 32, H, H, C, k, lib.F16, dw.data_ptr(), 32, st))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): lib.check(L.maf_dw_wgrad(x.data_ptr(), C, dy.data_ptr(), C, 32, H, H, C, k, lib.F16, dw.data_ptr(), 32, st))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('dw_wgrad %dx%d C=%d k=%d: %.1f us  (%.2f TB/s)' % (H, H, C, k, dt * 1e6, 2 * x.numel() * 2 / dt / 1e12))
