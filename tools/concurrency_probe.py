"""Do kernels of two HIP streams run concurrently on this device when each leaves most CUs idle?
    python tools/concurrency_probe.py
A chain of n small GroupNorm launches (16 workgroups, ~6 us each) on one stream vs two such chains on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops

dev = "cuda"
n = 200
def chain(x, g, b):
    for _ in range(n):
        ops.groupnorm_nhwc(x, g, b, 32, 1e-5, True)
xs = [torch.randn(2, 8, 8, 1280, device=dev).half() for _ in range(2)]
g = torch.ones(1280, device=dev); b = torch.zeros(1280, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two):
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): chain(xs[0], g, b)
    if two:
        with torch.cuda.stream(s2): chain(xs[1], g, b)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3
for _ in range(2): run(True)
one = min(run(False) for _ in range(3)); two = min(run(True) for _ in range(3))
print(f"eager: one chain of {n}: {one:.0f} us ({one / n:.2f} us per launch); two chains on two streams: {two:.0f} us  -> ratio {two / one:.2f}")
# the same under graph capture (no host in the way)
def graphed(two):
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): chain(xs[0], g, b)
        if two:
            with torch.cuda.stream(s2): chain(xs[1], g, b)
        cur.wait_stream(s1); cur.wait_stream(s2)
    g_.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g_.replay(); e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3
one, two = graphed(False), graphed(True)
print(f"graph: one chain: {one:.0f} us ({one / n:.2f} us per launch); two chains: {two:.0f} us -> ratio {two / one:.2f}")
