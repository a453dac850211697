"""Build libsdeo.so in-tree with hipcc for gfx950 (explicit `hipcc -shared -fPIC`, no JIT cache).

    python -m stablediffusioneo_amd.build [--force] [--debug]

--debug additionally builds libsdeo_dbg.so with -DSDEO_DEBUG_KERNELS: the conv / GEMM kernels then honour the SDEO_DBG_GEMM
ablation and stamp switches (tools/stamps.py, tools/halo_ablate.py; select it with SDEO_LIB=.../libsdeo_dbg.so).  The
production library contains none of those branches.

Objects are compiled one translation unit at a time (in parallel) into csrc/_build/ and linked into
stablediffusioneo_amd/libsdeo.so, which travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsdeo.so")
OBJ = os.path.join(CSRC, "_build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "sdeo.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, debug: bool = False) -> str:
    """debug: the measurement build (libsdeo_dbg.so)"""
    out = OUT.replace("libsdeo.so", "libsdeo_dbg.so") if debug else OUT
    objdir = OBJ + "_dbg" if debug else OBJ
    flags = FLAGS + (["-DSDEO_DEBUG_KERNELS"] if debug else [])
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "stamp")
    dig = _digest(CSRC) + ("-dbg" if debug else "")
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    srcs = _sources()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-6000:])
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {out} from {len(srcs)} translation units")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--debug" in sys.argv:
        build(force="--force" in sys.argv, debug=True)
