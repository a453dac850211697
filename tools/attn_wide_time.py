"""Device time of the wide-head attention (VAE AttnBlock shape): python tools/attn_wide_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
for (t, d) in [(4096, 512), (1024, 512), (4096, 256)]:
    q, k, v = (torch.randn(1, t, d, device="cuda").half() for _ in range(3))
    for _ in range(3): ops.attention(q, k, v, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.attention(q, k, v, 1)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"T{t} d{d}: {us:.0f} us  {4.0 * t * t * d / us / 1e6:.0f} TFLOP/s")
