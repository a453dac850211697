"""Sweep tile configuration x split-K for every distinct conv/GEMM problem of the configured networks
(GPU box only).  Device time per launch is taken from a replayed hipGraph of REP launches, so host launch
overhead does not pollute small kernels.  Prints, per shape, the heuristic's pick and the best measured plan.

    python tools/tune_gemm.py [--res 512] [--n 2]
"""
import argparse
import collections
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def census(res, n):
    code = (f"import sys; sys.path.insert(0, {ROOT!r});"
            "from stablediffusioneo_amd import spec as S; from stablediffusioneo_amd.runtime import SdeoRuntime;"
            "rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15);"
            f"rt.lib.sdeo_configure(rt.handle, {n}, {res // 8}, {res // 8})")
    env = dict(os.environ, SDEO_DUMP_GEMM="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stderr
    shapes = collections.Counter()
    for line in out.splitlines():
        if line.startswith("SDEO_GEMM"):
            f = line.split()
            shapes[tuple(int(v) for v in f[1:11]) + (f[11],)] += 1
    return shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--rep", type=int, default=10)
    a = ap.parse_args()
    import torch
    from stablediffusioneo_amd import _lib, ops
    lib = _lib.load()
    shapes = census(a.res, a.n)
    print(f"{len(shapes)} distinct problems, {sum(shapes.values())} launches (hint/ctx/cn/unet x2/vae programs)")
    dev = "cuda"
    total_h = total_b = 0.0
    rows = []
    for (M, N, K, Cin, R, stride, ups, B, Hi, Wi, name), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
        if Cin % 64:
            continue
        if R == 1:
            x = torch.randn(M, K, device=dev).half()
            w = (torch.randn(N, K, device=dev) * 0.02).half()
            fn = lambda: ops.gemm(x, w)
        else:
            x = torch.randn(B, Hi, Wi, Cin, device=dev).half()
            w = (torch.randn(N, R, R, Cin, device=dev) * 0.02).half()
            fn = lambda: ops.conv2d_nhwc(x, w, stride=stride, upsample2x=bool(ups))

        def timed():
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(a.rep):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) * 1e3 / (2 * a.rep)

        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
        t_h = timed()
        best = (t_h, "heur", 0)
        nk = K // 64
        res_ = {}
        for tile in (0, 1, 2, 5):
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if sk > 1 and nk // sk < 4:
                    continue
                lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
                try:
                    t = timed()
                except Exception:
                    continue
                res_[(tile, sk)] = t
                if t < best[0]:
                    best = (t, tile, sk)
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
        fl = 2.0 * M * N * K
        total_h += t_h * cnt
        total_b += best[0] * cnt
        per_tile = " ".join(f"t{t}:" + "/".join(f"{sk}={res_[(t, sk)]:.0f}" for sk in (1, 2, 3, 4, 6, 8, 12, 16) if (t, sk) in res_)
                            for t in (0, 1, 2, 5))
        print(f"M={M:6d} N={N:5d} K={K:6d} R={R} s{stride} u{ups} x{cnt:3d} heur[{name[21:-1]}]={t_h:7.1f}us ({fl / t_h / 1e6:6.0f}TF) "
              f"best=tile{best[1]} sk{best[2]} {best[0]:7.1f}us ({fl / best[0] / 1e6:6.0f}TF) | {per_tile}", flush=True)
    print(f"total heuristic {total_h / 1e3:.2f} ms, total best {total_b / 1e3:.2f} ms (all programs, one pass each)")


if __name__ == "__main__":
    main()
