"""Which operand's coldness costs a conv / GEMM launch its ~5 us (tools/cold_ab.py: 8 us hot -> 13 us behind a cache flush)?
After a 640 MB fill, touch (read) the weights and / or the activations with a torch reduction, then time ONE launch.

    python tools/cold_parts.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import ops                   # noqa: E402

dev = "cuda"
flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)


def timed(fn, pre, reps=9):
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        pre()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for (m, n, k) in [(8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (128, 1280, 1280), (512, 1280, 5120)]:
    x = torch.randn(m, k, device=dev).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    res = torch.randn(m, n, device=dev).half()
    fn = lambda: ops.gemm(x, w, res=res)
    fn(); fn()
    none = timed(fn, lambda: None)
    tw = timed(fn, lambda: w.float().sum())
    tx = timed(fn, lambda: (x.float().sum(), res.float().sum()))
    tb = timed(fn, lambda: (w.float().sum(), x.float().sum(), res.float().sum()))
    hot = timed(fn, lambda: (fn(), fn()))
    print(f"gemm M{m} N{n} K{k}: all cold {none:5.1f} us | weights touched {tw:5.1f} | activations + residual touched {tx:5.1f} | both {tb:5.1f} | "
          f"after two launches of itself {hot:5.1f}", flush=True)
for (nb, h, w_, cin, cout) in [(2, 8, 8, 1280, 1280), (2, 64, 64, 320, 320)]:
    x = torch.randn(nb, h, w_, cin, device=dev).half()
    wt = (torch.randn(cout, 3, 3, cin, device=dev) * (9 * cin) ** -0.5).half()
    fn = lambda: ops.conv2d_nhwc(x, wt)
    fn(); fn()
    none = timed(fn, lambda: None)
    tw = timed(fn, lambda: wt.float().sum())
    tx = timed(fn, lambda: x.float().sum())
    tb = timed(fn, lambda: (wt.float().sum(), x.float().sum()))
    hot = timed(fn, lambda: (fn(), fn()))
    print(f"conv3x3 {cin}->{cout} @{h}x{w_}: all cold {none:5.1f} us | weights touched {tw:5.1f} | activations touched {tx:5.1f} | both {tb:5.1f} | "
          f"after two launches of itself {hot:5.1f}", flush=True)
