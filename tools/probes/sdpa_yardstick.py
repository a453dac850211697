"""Yardstick only (not a product path): torch scaled_dot_product_attention (its ROCm flash / efficient back ends) on the step's
attention shapes, hipGraph of 20 launches:  python tools/probes/sdpa_yardstick.py"""
import torch, torch.nn.functional as F
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
from torch.nn.attention import sdpa_kernel, SDPBackend
for tq, tk, d in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80), (256, 77, 160)]:
    q = torch.randn(2, 8, tq, d, device="cuda").half(); k = torch.randn(2, 8, tk, d, device="cuda").half(); v = torch.randn(2, 8, tk, d, device="cuda").half()
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
        try:
            with sdpa_kernel(be):
                us = t(lambda: F.scaled_dot_product_attention(q, k, v))
            print(f"Tq{tq} Tk{tk} d{d} {be.name:20s}: {us:7.1f} us {4.0 * 16 * tq * tk * d / us / 1e6:6.0f} TFLOP/s", flush=True)
        except Exception as e:
            print(f"Tq{tq} Tk{tk} d{d} {be.name}: failed {str(e)[:80]}", flush=True)
