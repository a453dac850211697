import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops, _lib
lib = _lib.load()
x = torch.randn(64, device="cuda").half(); y = torch.empty_like(x)
xb = torch.randn(8192, 320, device="cuda").half(); g = torch.ones(320, device="cuda"); b = torch.zeros(320, device="cuda")
def t(fn, n=200):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("trivial silu(64 elems) per launch us:", t(lambda: lib.sdeo_debug_silu(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_int64(64), st())))
print("torch add (64 elems) per launch us:", t(lambda: torch.add(x, 1.0, out=y)))
print("layernorm 8192x320 us:", t(lambda: ops.layernorm(xb, g, b)))
# eager (no graph) back-to-back
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): lib.sdeo_debug_silu(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_int64(64), st())
torch.cuda.synchronize(); print("eager trivial per launch us:", (time.perf_counter() - t0) / 2000 * 1e6)
