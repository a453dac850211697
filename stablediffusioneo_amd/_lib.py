"""ctypes binding of libsdeo.so (the C ABI in include/sdeo.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, loading
raises immediately (the reference silently falls back to PyTorch when a .plan is missing,
`cldm_trt/ddim_hacked.py:22-23,35-36`; we deliberately do not)."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDEO_LIB") or os.path.join(HERE, "libsdeo.so")     # SDEO_LIB: A/B of two builds on one device
HEADER = os.path.join(os.path.dirname(HERE), "include", "sdeo.h")

_lib = None

MAX_LEVELS = 8


class SdeoConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("hint_channels", C.c_int),
        ("model_channels", C.c_int), ("num_res_blocks", C.c_int),
        ("channel_mult", C.c_int * MAX_LEVELS), ("num_levels", C.c_int),
        ("attention_resolutions", C.c_int * MAX_LEVELS), ("num_attention_resolutions", C.c_int),
        ("num_heads", C.c_int), ("context_dim", C.c_int), ("context_len", C.c_int),
        ("vae_ch", C.c_int), ("vae_out_ch", C.c_int), ("vae_ch_mult", C.c_int * MAX_LEVELS),
        ("vae_num_levels", C.c_int), ("vae_num_res_blocks", C.c_int), ("vae_z_channels", C.c_int),
        ("vae_scale_factor", C.c_float),
    ]


class SdeoClipConfig(C.Structure):
    _fields_ = [("vocab", C.c_int), ("positions", C.c_int), ("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
                ("ffn", C.c_int)]


def declared_symbols(header: str = HEADER):
    """Names of every function include/sdeo.h declares (used by the CPU export test)."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdeo_[a-z0-9_]+)\s*\(", txt)))


class SdeoError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SdeoError(f"{path} not found: build it with `python -m stablediffusioneo_amd.build` "
                        f"(there is no CPU / PyTorch fallback for the HIP path)")
    # Load order: PyTorch ships its own libamdhip64 / libhsa-runtime64 and libsdeo.so is linked against /opt/rocm's.  Whichever
    # is loaded first owns the soname; with libsdeo first, torch ends up with two HSA runtimes in the process and every later
    # HIP call fails with "no ROCm-capable device".  This host uses torch for device memory and streams, so torch goes first.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise SdeoError(f"libsdeo.so does not export {missing}")
    want = int(re.search(r"#define\s+SDEO_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    if lib.sdeo_version() != want:
        raise SdeoError(f"{path} reports ABI version {lib.sdeo_version()}, include/sdeo.h declares {want}: rebuild the library "
                        f"(operand semantics changed between versions)")
    lib.sdeo_last_error.restype = C.c_char_p
    for name in ("sdeo_groupnorm_workspace_bytes", "sdeo_conv2d_workspace_bytes", "sdeo_gemm_workspace_bytes",
                 "sdeo_device_bytes", "sdeo_clip_device_bytes"):
        getattr(lib, name).restype = C.c_size_t
    lib.sdeo_device_bytes.argtypes = [C.c_void_p]
    lib.sdeo_clip_device_bytes.argtypes = [C.c_void_p]
    lib.sdeo_tuned_gemm_plans_json.restype = C.c_char_p
    _lib = lib
    load_tuned_plans(lib)
    return lib


TUNED_PLANS = os.path.join(HERE, "tuned_plans_gfx950.json")


def load_tuned_plans(lib, path: str = TUNED_PLANS) -> int:
    """Push the committed (tile, split-K) table into the library; returns the number of entries."""
    import json
    env = os.environ.get("SDEO_TUNED_PLANS", "1")      # 0: re-measure everything (tools/tune_plans.py); a path: that table instead
    if env not in ("0", "1"):
        path = env
    if not os.path.exists(path) or env == "0":
        return 0
    rows = json.load(open(path))
    for r in rows:
        lib.sdeo_set_tuned_gemm_plan((C.c_int * 10)(*r[:10]), C.c_int(r[10]), C.c_int(r[11]))
    return len(rows)


def dump_tuned_plans(lib=None):
    import json
    lib = lib or load()
    return json.loads(lib.sdeo_tuned_gemm_plans_json().decode())


def check(rc: int, what: str = "sdeo"):
    if rc != 0:
        msg = load().sdeo_last_error().decode(errors="replace")
        raise SdeoError(f"{what} failed: {msg}")


def ptr(t):
    """Raw device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
