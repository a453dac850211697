"""Do kernels of two HIP streams overlap on this device?  tools/concurrency_probe.py could not tell: its eager form enqueued chain B only
after the host had finished enqueuing chain A (the host was the bottleneck), and its graph form depends on how hipGraph maps branches to
queues.  Here each kernel is LONG (a conv3x3 1280->1280 @8x8 forced onto a 40-workgroup plan: ~70 us on 16 % of the CUs) and the two
chains are enqueued INTERLEAVED (A1, B1, A2, B2, ...), so the host is never the limit.

    python tools/concurrency_probe2.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import _lib, ops             # noqa: E402

lib = _lib.load()
dev = "cuda"
n = 30
xs = [torch.randn(2, 8, 8, 1280, device=dev).half() for _ in range(2)]
ws = [(torch.randn(1280, 3, 3, 1280, device=dev) * 0.01).half() for _ in range(2)]
lib.sdeo_debug_force_gemm_plan(C.c_int(2), C.c_int(1))          # <64,64,4>, no split-K: 2 x 20 workgroups, 180 K-steps
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def enqueue(two):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(n):
        with torch.cuda.stream(s1):
            ops.conv2d_nhwc(xs[0], ws[0])
        if two:
            with torch.cuda.stream(s2):
                ops.conv2d_nhwc(xs[1], ws[1])
    cur.wait_stream(s1); cur.wait_stream(s2)


def eager(two):
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); enqueue(two); e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3


for _ in range(2):
    eager(True)
one = min(eager(False) for _ in range(3))
two = min(eager(True) for _ in range(3))
print(f"eager, interleaved enqueue: one chain of {n}: {one:.0f} us ({one / n:.1f} us per launch); two chains on two streams: {two:.0f} us -> ratio {two / one:.2f}", flush=True)


def graphed(two):
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        enqueue(two)
    g_.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e12
    for _ in range(3):
        a.record(); g_.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3)
    return best


one, two = graphed(False), graphed(True)
print(f"hipGraph replay: one chain: {one:.0f} us ({one / n:.1f} us per launch); two branches: {two:.0f} us -> ratio {two / one:.2f}", flush=True)
lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
