"""Streaming write / copy bandwidth of the device for tensors of the sizes the GEMM epilogues write (torch fill_ / copy_ kernels):
    python tools/bw_probe.py"""
import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for mb in (5, 21, 42, 168, 672):
    n = mb * 1024 * 1024 // 2
    y = torch.empty(n, dtype=torch.float16, device="cuda"); x = torch.randn(n, device="cuda").half()
    us = t(lambda: y.fill_(1.0)); print(f"fill {mb:4d} MB: {us:7.1f} us  {mb * 1.048576 / us * 1e3:6.0f} GB/s written")
    us = t(lambda: y.copy_(x)); print(f"copy {mb:4d} MB: {us:7.1f} us  {mb * 1.048576 / us * 1e3:6.0f} GB/s written (+ as much read)")
