"""Hot vs cold ranking of conv / GEMM plans.  The plan table was measured with 8 back-to-back launches of ONE problem, i.e. with its
weights and activations resident in L2 / Infinity Cache; inside a DDIM step every launch streams its weights from HBM (2.4 GB per
step pass through a 256 MB cache).  For each shape and candidate (tile, split-K):
   hot  = device time per launch of a hipGraph of 20 launches,
   cold = median of 7 single launches, each behind a 640 MB fill that evicts both cache levels (HIP events around the launch).

    python tools/cold_ab.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import _lib, ops             # noqa: E402

lib = _lib.load()
dev = "cuda"
flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)


def hot_us(fn, n=20):
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
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def cold_us(fn, reps=7):
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


GEMMS = [(8192, 320, 320), (8192, 320, 1280), (2048, 640, 640), (2048, 640, 2560), (512, 1280, 1280), (512, 1280, 5120), (128, 1280, 1280),
         (8192, 960, 320), (2048, 1920, 640), (512, 3840, 1280)]
CONVS = [(2, 8, 8, 1280, 1280), (2, 16, 16, 1280, 1280), (2, 64, 64, 320, 320), (2, 32, 32, 640, 640)]
DMA = [0, 1, 2, 5, 6, 7, 9, 10, 11, 12, 19, 20, 21, 26, 27, 28, 29, 30, 31, 32, 33, 41, 42, 43, 44, 45, 46, 47, 48]
HALO = [13, 14, 15, 16, 22, 23, 34, 35, 37, 38]
name = lib.sdeo_debug_conv2d_kernel_name
name.restype = C.c_char_p


def run(label, fn, tiles, sks):
    print(f"--- {label}", flush=True)
    rows = []
    for tile in [-1] + tiles:
        for sk in ([0] if tile < 0 else sks):
            lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
            try:
                try:
                    h = hot_us(fn)
                except Exception:
                    continue
                c = cold_us(fn)
                rows.append((c, h, tile, sk))
            finally:
                lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
    base = [r for r in rows if r[2] < 0][0]
    rows.sort()
    print(f"  tuned plan: cold {base[0]:.1f} us  hot {base[1]:.1f} us")
    for c, h, tile, sk in rows[:8]:
        print(f"  tile {tile:3d} sk {sk:2d}: cold {c:6.1f} us   hot {h:6.1f} us")
    sys.stdout.flush()


for (m, n, k) in GEMMS:
    x = torch.randn(m, k, device=dev).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    res = torch.randn(m, n, device=dev).half()
    run(f"gemm M{m} N{n} K{k} (+residual)", lambda: ops.gemm(x, w, res=res), DMA, [1, 2, 4, 8] if m <= 512 else [1, 2])
for (n, h, w_, cin, cout) in CONVS:
    x = torch.randn(n, h, w_, cin, device=dev).half()
    wt = (torch.randn(cout, 3, 3, cin, device=dev) * (9 * cin) ** -0.5).half()
    sks = [1, 2, 4, 8, 16] if n * h * w_ <= 512 else [1, 2]
    run(f"conv3x3 {cin}->{cout} @{h}x{w_} N={n}", lambda: ops.conv2d_nhwc(x, wt), DMA + HALO, sks)
