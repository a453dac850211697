"""A/B of two library builds on one device (SDEO_LIB), attention shapes of config 2: prints us per launch (hipGraph replay)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from stablediffusioneo_amd import ops
    from tools.bench_ops import timeit
    out = []
    for (t, tk, d) in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (4096, 77, 40)]:
        c = 8 * d
        tks = (tk + 7) // 8 * 8
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        q = (torch.randn(2, t, c, device="cuda", generator=g) * 0.5).half()
        k = (torch.randn(2, tks, c, device="cuda", generator=g) * 0.5).half()
        vt = torch.randn(2, tks, c, device="cuda", generator=g).half()
        us = min(timeit(lambda: ops.attention(q, k, vt, 8, tk=tk), iters=10) for _ in range(3))
        out.append(f"T{t}/Tk{tk}/d{d}={us:7.1f}us")
    print(f"{os.environ.get('SDEO_LIB', 'in-tree'):>40s}: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SDEO_LIB=lib), check=False)
