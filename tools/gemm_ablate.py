"""Ablations of one GEMM launch on the measurement build (libsdeo_dbg.so): which leg bounds a many-tile short-K GEMM?
    SDEO_LIB=.../libsdeo_dbg.so python tools/gemm_ablate.py M N K tile
Runs the shape under SDEO_DBG_GEMM = 0 (full), 32 (no epilogue), 4 (no MFMAs), 3 (both operands read the zero page: no L2 traffic
worth the name), 8 (no DMAs after the prologue), 16 (no fragment reads) in child processes (the switch is read once per process)."""
import os, subprocess, sys
if len(sys.argv) > 5:
    import ctypes as C, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stablediffusioneo_amd import ops, _lib
    lib = _lib.load()
    m, n, k, tile = (int(v) for v in sys.argv[1:5])
    x = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half(); b = torch.randn(n, device="cuda")
    lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(1))
    fn = lambda: ops.gemm(x, w, b, None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    print(f"{a.elapsed_time(e) / 40 * 1e3:.1f}")
else:
    for flag, what in [(0, "full"), (32, "no epilogue"), (4, "no MFMAs"), (3, "operands from the zero page"), (8, "no DMAs after the prologue"),
                       (16, "no fragment reads"), (36, "no MFMAs, no epilogue"), (35, "zero page + no epilogue"),
                       (512, "epilogue without its global stores"), (1024, "epilogue: staging only"), (2048 + 512, "epilogue: no staging writes, no stores"),
                       (2048 + 1024, "epilogue: set-up only")]:
        env = dict(os.environ, SDEO_DBG_GEMM=str(flag))
        r = subprocess.run([sys.executable, __file__] + sys.argv[1:5] + ["child"], env=env, capture_output=True, text=True)
        print(f"M{sys.argv[1]} N{sys.argv[2]} K{sys.argv[3]} tile {sys.argv[4]}  dbg {flag:2d} ({what}): {r.stdout.strip() or r.stderr[-200:]} us", flush=True)
