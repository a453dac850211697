"""Two chains of the SAME production-size kernel on two streams (interleaved enqueue) vs one chain: how much of a second stream's work
fits beside the first when a workgroup takes most of a CU's LDS (no co-residency) and when it takes less than half of it?

    python tools/concurrency_probe3.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import _lib, ops             # noqa: E402

lib = _lib.load()
dev = "cuda"
n = 40
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def graphed(fns, two):
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(n):
            with torch.cuda.stream(s1):
                fns[0]()
            if two:
                with torch.cuda.stream(s2):
                    fns[1]()
        cur.wait_stream(s1); cur.wait_stream(s2)
    g_.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e12
    for _ in range(3):
        a.record(); g_.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3)
    return best / n


def case(label, mk, tiles):
    fns = [mk(0), mk(1)]
    for tile, sk in tiles:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        try:
            fns[0](); fns[1]()
            one, two = graphed(fns, False), graphed(fns, True)
            print(f"{label} tile {tile:3d} sk {sk}: one chain {one:6.1f} us / launch; two chains {two:6.1f} us per PAIR -> ratio {two / one:.2f}", flush=True)
        except Exception as e:
            print(f"{label} tile {tile} sk {sk}: {str(e)[:80]}")
        finally:
            lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))


def gemm(m, n_, k):
    def mk(i):
        x = torch.randn(m, k, device=dev).half(); w = (torch.randn(n_, k, device=dev) * k ** -0.5).half(); r = torch.randn(m, n_, device=dev).half()
        return lambda: ops.gemm(x, w, res=r)
    return mk


def conv(nb, h, w_, cin, cout):
    def mk(i):
        x = torch.randn(nb, h, w_, cin, device=dev).half(); wt = (torch.randn(cout, 3, 3, cin, device=dev) * (9 * cin) ** -0.5).half()
        return lambda: ops.conv2d_nhwc(x, wt)
    return mk


case("gemm M8192 N320 K320  ", gemm(8192, 320, 320), [(6, 1), (27, 1), (30, 1), (32, 1)])
case("gemm M2048 N640 K640  ", gemm(2048, 640, 640), [(9, 1), (31, 1), (27, 1), (32, 1)])
case("gemm M512 N1280 K1280 ", gemm(512, 1280, 1280), [(2, 1), (42, 1), (27, 1), (32, 1)])
case("gemm M8192 N2560 K320 ", gemm(8192, 2560, 320), [(28, 1), (26, 1), (0, 1)])
case("conv 320->320 @64x64  ", conv(2, 64, 64, 320, 320), [(35, 1), (34, 1), (13, 1), (6, 1), (30, 1), (27, 1)])
case("conv 640->640 @32x32  ", conv(2, 32, 32, 640, 640), [(37, 1), (15, 1), (9, 1), (31, 1), (27, 2)])
case("conv 1280->1280 @16x16", conv(2, 16, 16, 1280, 1280), [(34, 4), (37, 2), (2, 6), (27, 6)])
case("conv 1280->1280 @8x8  ", conv(2, 8, 8, 1280, 1280), [(37, 5), (2, 12), (27, 12)])
