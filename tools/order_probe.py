"""Tile-order probe: time each conv/GEMM shape with M-fastest (0) and N-fastest (1) tile order (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import _lib, ops
lib = _lib.load()
dev = "cuda"
def timed(fn, rep=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (2 * rep)
shapes = [(2,64,320,320,3),(2,64,640,320,3),(2,64,960,320,3),(2,32,640,640,3),(2,32,1280,640,3),(2,16,1280,1280,3),(2,16,2560,1280,3),(2,8,1280,1280,3),(2,8,2560,1280,3),
          (1,512,128,128,3),(1,256,256,256,3),(1,128,512,512,3)]
for (b,hw,cin,cout,k) in shapes:
    x = torch.randn(b,hw,hw,cin,device=dev).half(); w = (torch.randn(cout,k,k,cin,device=dev)*0.02).half()
    r = []
    for o in (0,1,-1):
        lib.sdeo_debug_force_gemm_order(C.c_int(o)); r.append(timed(lambda: ops.conv2d_nhwc(x,w)))
    print(f"conv {cin}->{cout}@{hw} N={b}: M-fast {r[0]:.1f} N-fast {r[1]:.1f} heuristic {r[2]:.1f}", flush=True)
for (m,n,k) in [(8192,320,320),(8192,2560,320),(8192,320,1280),(2048,640,640),(2048,5120,640),(2048,640,2560),(512,1280,1280),(512,10240,1280),(512,1280,5120),(128,10240,1280)]:
    x = torch.randn(m,k,device=dev).half(); w = (torch.randn(n,k,device=dev)*0.02).half()
    r = []
    for o in (0,1,-1):
        lib.sdeo_debug_force_gemm_order(C.c_int(o)); r.append(timed(lambda: ops.gemm(x,w)))
    print(f"gemm {m}x{n}x{k}: M-fast {r[0]:.1f} N-fast {r[1]:.1f} heuristic {r[2]:.1f}", flush=True)
