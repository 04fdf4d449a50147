"""What the vendor GEMM does on the K-heavy 1x1 shapes of the m / s forwards (torch.addmm -> hipBLASLt / rocBLAS): the bar for the hand-written kernels.
   python tools/gemm_probe.py   (GPU box)"""
import torch, time
shapes = [(32 * 6400, 960, 384), (32 * 6400, 768, 384), (32 * 6400, 768, 256), (32 * 1600, 1280, 384), (32 * 1600, 1152, 384), (32 * 1600, 960, 384),
          (32 * 400, 1536, 768), (32 * 6400, 640, 256), (32 * 6400, 576, 192), (32 * 6400, 192, 576), (32 * 1600, 3456, 192), (32 * 1600, 1728, 192), (32 * 6400, 864, 192)]
dev = "cuda"
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(K, N, device=dev, dtype=torch.float16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16)
    for _ in range(5): y = torch.addmm(b, x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y = torch.addmm(b, x, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("M %7d K %5d N %4d  %.4f ms  %7.0f TFLOP/s  %6.0f GB/s" % (M, K, N, ms, 2.0 * M * K * N / ms / 1e9, (M * K + M * N) * 2 / ms / 1e6), flush=True)
