"""Run one conv/GEMM problem repeatedly (for rocprofv3 --pmc): python tools/one_conv.py M N K R [tile sk reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import _lib, ops
a = [int(v) for v in sys.argv[1:]]
M, N, K, R = a[:4]
tile, sk, reps = (a[4:7] + [-1, 0, 20][len(a) - 4:]) if len(a) > 4 else (-1, 0, 20)
lib = _lib.load()
lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
dev = "cuda"
if R == 1:
    x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.02).half()
    fn = lambda: ops.gemm(x, w)
else:
    cin = K // (R * R); hw = int(round((M // 2) ** 0.5))
    x = torch.randn(2, hw, hw, cin, device=dev).half(); w = (torch.randn(N, R, R, cin, device=dev) * 0.02).half()
    fn = lambda: ops.conv2d_nhwc(x, w)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done")
