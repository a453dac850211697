"""Attention-only micro-benchmark (graph-timed). GPU box only. SDEO_ATTN_KS=1|2 forces the key-split variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
from tools.bench_ops import timeit, rnd  # noqa

for (t, tk, d) in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80),
                   (9216, 9216, 40), (2304, 2304, 80)]:
    c = 8 * d
    tks = (tk + 7) // 8 * 8
    q = rnd(2, t, c); k = rnd(2, tks, c); vt = rnd(2, tks, c)
    us = timeit(lambda: ops.attention(q, k, vt, 8, tk=tk))
    print(f"attn T={t:5d} Tk={tk:5d} d={d:3d}: {us:9.1f} us  {4.0*2*8*t*tk*d/us/1e6:8.1f} TFLOP/s", flush=True)
