"""Which SIMD each wave of a 512-thread workgroup lands on (HW_REG_HW_ID), halo kernel.  SDEO_DBG_GEMM=256."""
import ctypes as C, os, sys, collections
os.environ["SDEO_DBG_GEMM"] = "256"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stablediffusioneo_amd import _lib, ops
from tools.bench_ops import rnd
lib = _lib.load()
x = rnd(2, 64, 64, 320); w = rnd(320, 3, 3, 320, scale=0.02); b = torch.zeros(320, device="cuda")
lib.sdeo_debug_force_gemm_plan(C.c_int(13), C.c_int(1))
ops.conv2d_nhwc(x, w, b); torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.sdeo_debug_read_stamps(C.c_int(1), buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
hw = buf.reshape(4096, 16)[:256, :8].astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
wid = hw & 15
pat = collections.Counter(tuple(r) for r in simd)
print("SIMD of waves 0..7, by frequency over 256 workgroups:")
for k, v in pat.most_common(12):
    print("  ", k, v)
print("first workgroups raw (wave_id, simd, cu):")
for i in range(4):
    print("  ", [(int(wid[i, j]), int(simd[i, j]), int(cu[i, j])) for j in range(8)])
