"""Traffic ablation of the implicit-GEMM kernel (measurement only): SDEO_DBG_GEMM=0|1|2|3 python tools/gemm_ablate.py
bit 0: activation DMAs read the zero page (no L2 traffic for B), bit 1: weight DMAs do (no L2 traffic for A)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import _lib, ops
from tools.bench_ops import timeit, rnd
lib = _lib.load()
print("SDEO_DBG_GEMM =", os.environ.get("SDEO_DBG_GEMM", "0"))
CASES = [(8192, 320, 2880, 3, [(6, 1), (7, 1)]),
         (8192, 640, 5760, 3, [(7, 1)])]
for (M, N, K, R, plans) in CASES:
    if R == 1:
        x = rnd(M, K); w = rnd(N, K, scale=0.02)
        fn = lambda: ops.gemm(x, w)
    else:
        cin = K // 9; hw = int(round((M // 2) ** 0.5))
        x = rnd(2, hw, hw, cin); w = rnd(N, 3, 3, cin, scale=0.02)
        fn = lambda: ops.conv2d_nhwc(x, w)
    out = []
    for (tile, sk) in plans:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        us = timeit(fn)
        out.append(f"t{tile}/sk{sk}={us:6.1f}us({2.0*M*N*K/us/1e6:4.0f}TF)")
    print(f"M={M:6d} N={N:5d} K={K:6d} R={R}: " + "  ".join(out), flush=True)
