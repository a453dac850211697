"""Where a conv / GEMM launch spends its time: per-workgroup phase stamps (SDEO_DBG_GEMM=64, measurement only).

    SDEO_DBG_GEMM=64 python tools/stamps.py

For each case: the launch is replayed a few times back to back; the LAST launch's stamps are read.  All times in us relative
to the earliest workgroup entry: entry (dispatch ramp), set-up done, first K-step visible, loop done, stores issued, stores
complete; min / median / max over workgroups."""
import ctypes as C, os, sys
os.environ.setdefault("SDEO_DBG_GEMM", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stablediffusioneo_amd import _lib, ops
from tools.bench_ops import timeit, rnd

lib = _lib.load()
NAMES = ["entry", "setup", "first data", "loop done", "stores issued", "stores done", "2nd epilogue", "-", "epi enter", "epi res/shfl", "epi staged", "epi blk0 stored", "epi all stored"]


def report(which, nwg, label, us):
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    lib.sdeo_debug_read_stamps(C.c_int(which), buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
    st = buf.reshape(4096, 16)[:nwg, :13].astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0
    full = buf.reshape(4096, 16)[:nwg].astype(np.int64)
    full = full[full[:, 0] > 0]
    if full[:, 15].max() > 0:
        ghz = (full[:, 15] - full[:, 14]) / np.maximum(full[:, 3] - full[:, 0], 1) * 0.1
        print(f"    shader clock over entry..loop done: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})")
    print(f"{label}: {us:.1f} us per launch back-to-back, {len(st)} workgroups")
    for i, n in enumerate(NAMES):
        c = rel[:, i]
        if c.max() <= 0 or n == "-":
            continue
        print(f"    {n:14s} min {c.min():7.2f}  med {np.median(c):7.2f}  max {c.max():7.2f}")


def main():
    for (n, cin, hw, cout, tile, sk, which) in [(2, 320, 64, 320, 13, 1, 1), (2, 320, 64, 320, 6, 1, 0), (2, 640, 32, 640, 13, 2, 1),
                                                (2, 1280, 16, 1280, 13, 4, 1)]:
        x = rnd(n, hw, hw, cin); w = rnd(cout, 3, 3, cin, scale=0.02); b = torch.zeros(cout, device="cuda")
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        us = timeit(lambda: ops.conv2d_nhwc(x, w, b), iters=10)
        ntile = {13: (n * (hw // 8) * (hw // 16)) * (cout // 80), 6: ((n * hw * hw + 63) // 64) * ((cout + 159) // 160)}[tile]
        report(which, min(4096, ntile * sk), f"conv3x3 {cin}->{cout} @{hw} tile {tile} sk {sk}", us)
    for (m, nn, k, tile) in [(8192, 320, 320, 6), (2048, 640, 640, 9), (512, 1280, 1280, 2), (8192, 2560, 320, 1)]:
        x = rnd(m, k); w = rnd(nn, k, scale=0.02); b = torch.zeros(nn, device="cuda")
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(1))
        us = timeit(lambda: ops.gemm(x, w, b), iters=10)
        bm, bn = {6: (64, 160), 9: (32, 160), 2: (64, 64), 1: (128, 64)}[tile]
        report(0, min(4096, ((m + bm - 1) // bm) * ((nn + bn - 1) // bn)), f"gemm {m}x{nn}x{k} tile {tile}", us)


if __name__ == "__main__":
    main()
