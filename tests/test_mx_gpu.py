"""Block-scaled fp8 ("MX") GEMM path (BASELINE configs[4]: "fp8 ... on CDNA4 fp8 MFMA"): both operands as OCP e4m3fn codes with one
e8m0 scale per 32 elements of a row, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/conv_gemm.hip MX).

  * the device pack (quantize_mx_kernel) is bit-identical to oracle/fp8_quant.py quantize_mx, which is pinned to torch's own
    float8_e4m3fn cast;
  * the GEMM equals the fp32 product of the DEQUANTISED operands (the hardware multiplies decoded codes and scales exactly and
    accumulates in fp32: what is left is summation order and the fp16 rounding of the output): rtol 1e-3 + atol 1e-3 of the output
    scale, on every MX tile, ragged M / N, bias + residual, split-K;
  * the distance to the UNQUANTISED product is printed (a property of the format: ~3 % per operand element, shrinking with K)."""
import ctypes as C

import pytest
import torch

from tests.common import randn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def h16(t):
    return t.to(torch.float16)


def test_mx_pack_matches_oracle():
    from oracle import fp8_quant as Q
    from stablediffusioneo_amd import ops
    g = torch.Generator().manual_seed(7)
    for rows, cols in ((64, 320), (37, 2880 - 2880 % 32), (300, 1280)):
        x = h16(torch.randn((rows, cols), generator=g) * torch.logspace(-3, 2, rows)[:, None])
        x[3] = 0
        x[5, :8] = torch.tensor([448, -448, 0.001, -0.0019, 2 ** -9 * 3, 2 ** -10, 65504, -1e-7], dtype=torch.float16)
        x[6, 32:64] = 0                                   # an all-zero block inside a non-zero row
        q, sc = ops.quantize_mx(x.to(DEV))
        q0, sc0, _ = Q.quantize_mx(x)
        assert torch.equal(sc.cpu(), sc0), f"scales differ in {int((sc.cpu() != sc0).sum())} blocks"
        assert torch.equal(q.cpu(), q0), f"codes differ in {int((q.cpu() != q0).sum())} places"


MX_TILES = [0, 1, 6, 21]


@pytest.mark.parametrize("tile", MX_TILES)
@pytest.mark.parametrize("shape", [(256, 320, 128, 1), (1000, 328, 640, 1), (8192, 320, 1280, 1), (512, 1280, 2560, 2), (130, 72, 384, 1)])
def test_mx_gemm_vs_dequantised_product(tile, shape):
    from oracle import fp8_quant as Q
    from stablediffusioneo_amd import _lib, ops
    m, n, k, sk = shape
    lib = _lib.load()
    x = h16(randn((m, k), 70 + k) * torch.logspace(-1, 1, m)[:, None])          # rows of very different magnitude: the block scales matter
    w = h16(randn((n, k), 71) * k ** -0.5 * torch.logspace(-1, 0.5, n)[:, None])
    w[:, 64:96] *= 40.0                                                          # ... and so do blocks inside a row
    bias = 0.1 * randn((n,), 72)
    res = h16(randn((m, n), 73))
    xq, xs = ops.quantize_mx(x.to(DEV))
    wq, ws = ops.quantize_mx(w.to(DEV))
    try:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        y = ops.gemm_mx(xq, xs, wq, ws, bias=bias.to(DEV), res=res.to(DEV))
    finally:
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
    xd, wd = Q.quantize_mx(x)[2], Q.quantize_mx(w)[2]
    ref = xd @ wd.t() + bias + res.float()
    scale = float(ref.abs().max())
    err = (y.float().cpu() - ref).abs()
    assert bool((err <= 1e-3 * ref.abs() + 1e-3 * scale).all()), f"tile {tile} {shape}: max err {float(err.max()):.3e} (scale {scale:.3g})"
    full = x.float() @ w.float().t() + bias + res.float()
    print(f"[mx] tile {tile} {shape}: max|err| vs dequantised product {float(err.max()) / scale:.2e} of scale; "
          f"quantisation moves the product by {float((ref - full).abs().max()) / scale:.2e} of scale")
