"""Ablation of the halo 3x3 kernel and the implicit-GEMM kernel on the model's 3x3 shapes (measurement only; results are wrong
under a non-zero flag).  One process per flag value because the library reads SDEO_DBG_GEMM once:

    python tools/halo_ablate.py            # runs itself for SDEO_DBG_GEMM in 0, 4 (no MFMA), 8 (no DMA after the prologue),
                                           # 16 (no fragment reads), 12, 20, 28, 32 (no epilogue)
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # (n, cin, hw, cout, plans [(tile, sk)])
    (2, 320, 64, 320, [(13, 1), (22, 1), (6, 1)]),
    (2, 640, 32, 640, [(13, 2), (22, 2), (15, 1)]),
    (2, 1280, 16, 1280, [(13, 4), (22, 4)]),
]


def child():
    import torch
    from stablediffusioneo_amd import _lib, ops
    from tools.bench_ops import timeit, rnd
    lib = _lib.load()
    out = []
    for (n, cin, hw, cout, plans) in CASES:
        x = rnd(n, hw, hw, cin); w = rnd(cout, 3, 3, cin, scale=0.02); b = torch.zeros(cout, device="cuda")
        fl = 2.0 * n * hw * hw * cout * 9 * cin
        for (tile, sk) in plans:
            lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
            us = timeit(lambda: ops.conv2d_nhwc(x, w, b), iters=20)
            out.append(f"c{cin}@{hw} t{tile}/sk{sk}={us:6.1f}us")
    print(f"DBG={os.environ.get('SDEO_DBG_GEMM', '0'):>3s}: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for flag in (0, 4, 8, 16, 28, 32, 60):
            env = dict(os.environ, SDEO_DBG_GEMM=str(flag))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
