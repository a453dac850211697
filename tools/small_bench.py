"""Graph-timed micro-benchmarks of the non-GEMM kernels at the shapes of config 2. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
from tools.bench_ops import timeit, rnd
dev = "cuda"
print("== groupnorm+silu (N=2)")
for (c, hw) in [(320, 64), (640, 64), (960, 64), (640, 32), (1920, 32), (1280, 16), (2560, 16), (1280, 8), (2560, 8)]:
    x = rnd(2, hw, hw, c); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    us = timeit(lambda: ops.groupnorm_nhwc(x, g, b, 32, 1e-5, True))
    by = 2 * hw * hw * c * 2 * 3
    print(f"gn C={c:5d} @{hw:3d}: {us:8.1f} us  {by/us/1e3:8.1f} GB/s (3 passes)", flush=True)
print("== layernorm / geglu")
for (rows, c) in [(8192, 320), (2048, 640), (512, 1280), (128, 1280)]:
    x = rnd(rows, c); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    us = timeit(lambda: ops.layernorm(x, g, b))
    print(f"ln {rows}x{c}: {us:8.1f} us {rows*c*4/us/1e3:8.1f} GB/s")
    a = rnd(rows, 8 * c)
    us = timeit(lambda: ops.geglu(a))
    print(f"geglu {rows}x{4*c}: {us:8.1f} us {rows*c*4*2*3/us/1e3:8.1f} GB/s", flush=True)
