"""Micro-benchmarks of the hand-written kernels on the shapes of config 2 (512x512, N=2). GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops

dev = "cuda"

def timeit(fn, iters=10, warm=2):
    """device time per call in us: replay of a hipGraph holding `iters` calls (no host launch overhead)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * iters) * 1e3   # us

def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).half()

def main():
    print("== conv3x3 (N=2)")
    for (cin, cout, hw, stride, ups) in [(320, 320, 64, 1, 0), (640, 320, 64, 1, 0), (960, 320, 64, 1, 0), (640, 640, 32, 1, 0),
                                         (1280, 640, 32, 1, 0), (1920, 640, 32, 1, 0), (1280, 1280, 16, 1, 0), (2560, 1280, 16, 1, 0),
                                         (1280, 1280, 8, 1, 0), (2560, 1280, 8, 1, 0), (320, 320, 64, 2, 0), (640, 640, 32, 1, 1),
                                         (1280, 1280, 16, 1, 1)]:
        x = rnd(2, hw, hw, cin); w = rnd(cout, 3, 3, cin, scale=0.02); b = torch.zeros(cout, device=dev)
        us = timeit(lambda: ops.conv2d_nhwc(x, w, b, stride=stride, upsample2x=bool(ups)))
        ho = hw * (2 if ups else 1) // stride
        fl = 2.0 * 2 * ho * ho * cout * 9 * cin
        print(f"conv3x3 {cin:5d}->{cout:5d} @{hw:3d} s{stride} u{ups}: {us:9.1f} us  {fl/us/1e6:8.1f} TFLOP/s")

    print("== conv 128ch VAE (N=1)")
    for (cin, cout, hw) in [(128, 128, 512), (256, 256, 256), (512, 512, 128), (512, 512, 64)]:
        x = rnd(1, hw, hw, cin); w = rnd(cout, 3, 3, cin, scale=0.02); b = torch.zeros(cout, device=dev)
        us = timeit(lambda: ops.conv2d_nhwc(x, w, b), iters=5)
        fl = 2.0 * hw * hw * cout * 9 * cin
        print(f"conv3x3 {cin:5d}->{cout:5d} @{hw:3d}: {us:9.1f} us  {fl/us/1e6:8.1f} TFLOP/s")

    print("== gemm")
    for (m, n, k) in [(8192, 320, 320), (8192, 2560, 320), (8192, 320, 1280), (2048, 640, 640), (2048, 5120, 640), (2048, 640, 2560),
                      (512, 1280, 1280), (512, 10240, 1280), (512, 1280, 5120), (128, 1280, 1280), (128, 10240, 1280), (160, 1280, 768),
                      (2, 1280, 1280)]:
        x = rnd(m, k); w = rnd(n, k, scale=0.02); b = torch.zeros(n, device=dev)
        us = timeit(lambda: ops.gemm(x, w, b))
        print(f"gemm {m:5d}x{n:5d}x{k:5d}: {us:9.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s   {(n*k*2)/us/1e3:8.1f} GB/s(w)")

    print("== attention (B=2, H=8)")
    for (t, tk, d) in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80)]:
        c = 8 * d
        tks = (tk + 7) // 8 * 8
        q = rnd(2, t, c); k = rnd(2, tks, c); vt = rnd(2, tks, c)
        us = timeit(lambda: ops.attention(q, k, vt, 8, tk=tk))
        print(f"attn T={t:5d} Tk={tk:5d} d={d:3d}: {us:9.1f} us  {4.0*2*8*t*tk*d/us/1e6:8.1f} TFLOP/s")

    print("== groupnorm+silu (N=2)")
    for (c, hw) in [(320, 64), (640, 64), (960, 64), (640, 32), (1920, 32), (1280, 16), (2560, 16), (1280, 8), (2560, 8)]:
        x = rnd(2, hw, hw, c); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
        us = timeit(lambda: ops.groupnorm_nhwc(x, g, b, 32, 1e-5, True))
        by = 2 * hw * hw * c * 2 * 3
        print(f"gn C={c:5d} @{hw:3d}: {us:8.1f} us  {by/us/1e3:8.1f} GB/s (3 passes)")
    print("== layernorm / geglu")
    for (rows, c) in [(8192, 320), (2048, 640), (512, 1280)]:
        x = rnd(rows, c); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
        us = timeit(lambda: ops.layernorm(x, g, b))
        print(f"ln {rows}x{c}: {us:8.1f} us {rows*c*4/us/1e3:8.1f} GB/s")
        a = rnd(rows, 8 * c)
        us = timeit(lambda: ops.geglu(a))
        print(f"geglu {rows}x{4*c}: {us:8.1f} us {rows*c*4*2*3/us/1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
