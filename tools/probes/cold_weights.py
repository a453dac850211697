"""How much of a launch's in-step time is the COLD WEIGHT STREAM?  Inside a DDIM step every conv / GEMM reads its weights from HBM
(2.4 GB per step through a 256 MB Infinity Cache) while its activations were written microseconds earlier.  Per shape, device time
per launch of one hipGraph of R launches on the same activations with
   hot:   one weight matrix (resident in L2 / Infinity Cache after the first launch),
   cold:  R distinct copies of the weight matrix, R x bytes > 700 MB where R <= 400 allows it (each launch streams from HBM),
   mall:  R copies with R x bytes ~ 100 MB (beyond the 32 MB of L2, inside the Infinity Cache).
Measurement only.   python tools/probes/cold_weights.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import ops                   # noqa: E402

dev = "cuda"


def graph_us(fns):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / len(fns) * 1e3)
    return best


def copies(w, total_bytes, cap=400):
    r = max(2, min(cap, int(total_bytes // (w.numel() * 2)) + 1))
    return [w.clone() for _ in range(r)]


def gemm_case(m, n, k):
    x = (torch.randn(m, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    b = torch.zeros(n, device=dev)
    hot = graph_us([lambda: ops.gemm(x, w, bias=b)] * 40)
    wc = copies(w, 700 << 20)
    cold = graph_us([(lambda wi=wi: ops.gemm(x, wi, bias=b)) for wi in wc])
    wm = copies(w, 100 << 20)
    mall = graph_us([(lambda wi=wi: ops.gemm(x, wi, bias=b)) for wi in wm])
    print(f"gemm M{m} N{n} K{k} ({n * k * 2 / 1e6:.1f} MB of weights): hot {hot:6.1f} us   mall {mall:6.1f} us ({len(wm)} copies)   "
          f"cold {cold:6.1f} us ({len(wc)} copies, {len(wc) * n * k * 2 / 1e6:.0f} MB)", flush=True)


def conv_case(nb, h, wd, cin, cout):
    x = (torch.randn(nb, h, wd, cin, device=dev) * 0.5).half()
    w = (torch.randn(cout, 3, 3, cin, device=dev) * (9 * cin) ** -0.5).half()
    b = torch.zeros(cout, device=dev)
    hot = graph_us([lambda: ops.conv2d_nhwc(x, w, bias=b)] * 40)
    wc = copies(w, 700 << 20)
    cold = graph_us([(lambda wi=wi: ops.conv2d_nhwc(x, wi, bias=b)) for wi in wc])
    wm = copies(w, 100 << 20)
    mall = graph_us([(lambda wi=wi: ops.conv2d_nhwc(x, wi, bias=b)) for wi in wm])
    print(f"conv3x3 {cin}->{cout} @{h}x{wd} N={nb} ({w.numel() * 2 / 1e6:.1f} MB of weights): hot {hot:6.1f} us   mall {mall:6.1f} us "
          f"({len(wm)} copies)   cold {cold:6.1f} us ({len(wc)} copies, {len(wc) * w.numel() * 2 / 1e6:.0f} MB)", flush=True)


for (m, n, k) in [(8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (128, 1280, 1280), (512, 10240, 1280), (512, 1280, 6400)]:
    gemm_case(m, n, k)
for c in [(2, 8, 8, 1280, 1280), (2, 16, 16, 1280, 1280), (2, 32, 32, 640, 640), (2, 64, 64, 320, 320)]:
    conv_case(*c)
