"""Device time of single GEMM shapes (back-to-back launches, HIP events):  python tools/gemm_time.py
   shapes: the short-K, many-tile GEMMs of the transformer blocks (ff.net.0.proj + GEGLU, q|k|v)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops

def t(fn, n=30):
    """n launches captured in one hipGraph (no host work between them), replayed three times"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3

dev = "cuda"
for (m, c) in [(8192, 320), (2048, 640), (512, 1280)]:
    x = torch.randn(m, c, device=dev).half()
    w = (torch.randn(8 * c, c, device=dev) * c ** -0.5).half()
    bias = torch.randn(8 * c, device=dev)
    wi, bi = ops.geglu_interleave(w), ops.geglu_interleave(bias)
    us = t(lambda: ops.gemm_geglu(x, wi, bi))
    fl = 2.0 * m * 8 * c * c
    print(f"ff1+GEGLU M{m} N{8*c} K{c}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s")
    for n in (8 * c, 3 * c, c):
        w2 = (torch.randn(n, c, device=dev) * c ** -0.5).half()
        us = t(lambda: ops.gemm(x, w2, None, None))
        print(f"gemm M{m} N{n} K{c}: {us:.1f} us  {2.0 * m * n * c / us / 1e6:.0f} TFLOP/s")
