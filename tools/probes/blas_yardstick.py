"""Yardstick only (not a product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the step's GEMM shapes, hipGraph of 20 launches.
    python tools/probes/blas_yardstick.py"""
import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3
shapes = [(8192, 2560, 320), (2048, 5120, 640), (512, 10240, 1280), (8192, 320, 1280), (2048, 640, 2560), (512, 1280, 5120), (512, 1280, 1280),
          (8192, 320, 320), (2048, 640, 640), (128, 1280, 1280), (8192, 960, 320), (2048, 1920, 640), (512, 3840, 1280), (32768, 1280, 1280),
          (8192, 320, 2880), (2048, 640, 5760), (512, 1280, 11520), (128, 1280, 11520)]
for m, n, k in shapes:
    x = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    y = torch.empty(m, n, device="cuda", dtype=torch.float16)
    us = t(lambda: torch.matmul(x, w.t(), out=y))
    print(f"M{m} N{n} K{k}: {us:7.1f} us {2.0 * m * n * k / us / 1e6:6.0f} TFLOP/s", flush=True)
