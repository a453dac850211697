"""Time one GEMM shape on every DMA tile (forced plan), n launches in a hipGraph:  python tools/gemm_tiles.py M N K [geglu]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops, _lib
lib = _lib.load()
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
geglu = len(sys.argv) > 4
x = torch.randn(m, k, device="cuda").half()
w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
bias = torch.randn(n, device="cuda")
names = {0: "128x128x3", 1: "128x64x3", 2: "64x64x4", 5: "256x128x3", 6: "64x160x3", 7: "128x160x3", 8: "256x160x3", 9: "32x160x4", 10: "64x160x5",
         11: "128x160x4", 12: "128x64x5", 19: "64x64x8", 20: "32x160x6", 21: "128x128x4", 26: "128x64x2 L", 27: "64x64x2 L", 28: "128x128x2 L", 29: "128x160x2 L", 30: "64x160x2 L", 31: "32x160x2 L", 32: "64x64x3 L", 33: "128x64x3 L"}
def t(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3
for tile, nm in names.items():
    if geglu and tile not in (0, 1, 2, 5, 12, 19, 21, 26, 27, 28, 32, 33): continue
    lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(1))
    try:
        if geglu:
            us = t(lambda: ops.gemm_geglu(x, w, bias))
        else:
            us = t(lambda: ops.gemm(x, w, bias, None))
        print(f"M{m} N{n} K{k} {'geglu ' if geglu else ''}tile {tile:2d} {nm:10s}: {us:6.1f} us  {2.0 * m * n * k / us / 1e6:5.0f} TFLOP/s")
    except Exception as e:
        print(tile, nm, "failed", str(e)[:80])
lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
