"""Same-device A/B of conv plans for the [conv3x3 -> GroupNorm] pairs of a ResBlock: device time (hipGraph of 20 chains, replayed) of
  (a) conv at its tuned plan + two-pass GroupNorm,  (b) conv at a forced (tile, split-K) + whichever GroupNorm form that plan allows
      (normalise-only when its epilogue can emit the partials, two-pass otherwise).

    python tools/plan_ab.py            # the 32x32 / 64x64 ResBlock shapes of SD-1.5 at N = 2
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import _lib, ops             # noqa: E402

lib = _lib.load()
dev = "cuda"


def graph_us(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


SHAPES = [(2, 64, 64, 320, 320), (2, 64, 64, 640, 320), (2, 32, 32, 640, 640), (2, 32, 32, 1280, 640), (2, 16, 16, 1280, 1280), (2, 8, 8, 1280, 1280),
          (1, 512, 512, 128, 128), (1, 256, 256, 256, 256)]
CANDS = [(-1, 0), (13, 1), (34, 1), (22, 1), (35, 1), (15, 1), (37, 1), (16, 1), (38, 1), (17, 1), (36, 1), (24, 1), (40, 1), (13, 2), (34, 2), (22, 2), (35, 2), (15, 2), (37, 2),
         (13, 4), (34, 4), (37, 4), (15, 4), (15, 8), (37, 8)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]

for (n, h, w, cin, cout) in SHAPES:
    x = (torch.randn(n, h, w, cin, device=dev)).half()
    wt = (torch.randn(cout, 3, 3, cin, device=dev) * (9 * cin) ** -0.5).half()
    bias = torch.randn(cout, device=dev)
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    print(f"--- conv3x3 {cin}->{cout} @{h}x{w} N={n}  (M {n * h * w}, K {9 * cin})", flush=True)
    for tile, sk in CANDS:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        try:
            name = lib.sdeo_debug_conv2d_kernel_name
            name.restype = C.c_char_p
            kn = name(C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin), C.c_int(cout), C.c_int(3), C.c_int(1), C.c_int(0)).decode()
            try:
                conv_us = graph_us(lambda: ops.conv2d_nhwc(x, wt, bias=bias))
            except Exception as e:          # plan not applicable to the shape
                print(f"  tile {tile:3d} sk {sk}: {str(e)[:80]}")
                continue
            emit = ops.conv2d_gn(x, wt, gamma, beta, bias=bias, swish=True)
            if emit is not None:
                chain = graph_us(lambda: ops.conv2d_gn(x, wt, gamma, beta, bias=bias, swish=True))
                form = f"emit+apply (slots {emit[2]})"
            else:
                def two():
                    y = ops.conv2d_nhwc(x, wt, bias=bias)
                    ops.groupnorm_nhwc(y, gamma, beta, 32, 1e-5, True)
                chain = graph_us(two)
                form = "two-pass GroupNorm"
            print(f"  tile {tile:3d} sk {sk}: {kn:42s} conv {conv_us:6.1f} us   conv+GN {chain:6.1f} us  [{form}]", flush=True)
        finally:
            lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
