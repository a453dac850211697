"""fp16 GEMM vs block-scaled fp8 GEMM (v_mfma_scale_f32_16x16x128_f8f6f4), device time per launch under hipGraph replay:
   fp16 | MX GEMM alone (operands already packed) | activation pack + MX GEMM (the pack as its own launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
dev = "cuda"
def graph_us(fn, n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best
for (m, n, k) in [(8192, 320, 1280), (8192, 1280, 1280), (8192, 2560, 1280), (8192, 2560, 640), (2048, 640, 2560), (2048, 5120, 640), (2048, 1920, 640), (32768, 1280, 1280), (512, 1280, 5120)]:
    x = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * k ** -0.5).half(); res = torch.randn(m, n, device=dev).half()
    xq, xs = ops.quantize_mx(x); wq, ws = ops.quantize_mx(w)
    t16 = graph_us(lambda: ops.gemm(x, w, res=res))
    tmx = graph_us(lambda: ops.gemm_mx(xq, xs, wq, ws, res=res))
    def both():
        a, b = ops.quantize_mx(x)
        ops.gemm_mx(a, b, wq, ws, res=res)
    tb = graph_us(both)
    fl = 2.0 * m * n * k
    print(f"M{m} N{n} K{k}: fp16 {t16:6.1f} us ({fl / t16 / 1e6:5.0f} TF)   MX {tmx:6.1f} us ({fl / tmx / 1e6:5.0f} TF)   pack + MX {tb:6.1f} us", flush=True)
