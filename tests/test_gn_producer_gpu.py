"""GroupNorm statistics out of the producing conv's epilogue (csrc/conv_inl.h KP::gn_out, csrc/norm.hip ext_partials): the
[conv -> GroupNorm] pairs of ResBlock / ResnetBlock / SpatialTransformer (`openaimodel.py:255-275`, `model.py:129-149`,
`attention.py:431-436`) run as [conv + partial (sum, sumsq) per (image, M tile, group)] -> [normalise only].  Checked against a
torch fp32 GroupNorm of the conv's own fp16 output (rtol 2e-3 + atol 3e-3, the bound of the two-pass kernel's test), against the
two-pass kernel, for bitwise run-to-run determinism, and that the conv's output bits do not depend on the emission."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.common import randn

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from stablediffusioneo_amd import ops as o
    return o


def h16(t):
    return t.half()


# (kTiles index, n, cin, h, w, cout, k, residual)  -- tile forced with split-K 1; cpg = cout / 32
CASES = [
    (13, 2, 64, 32, 32, 320, 3, True),      # halo <8,16,80,4>, cpg 10
    (22, 2, 128, 16, 32, 640, 3, False),    # halo <8,16,80,8>, cpg 20
    (14, 1, 64, 16, 16, 1280, 3, True),     # halo <8,16,160,4>, cpg 40
    (23, 2, 64, 8, 16, 2560, 3, False),     # halo <8,16,160,8>, cpg 80
    (25, 1, 64, 256, 128, 128, 3, True),    # halo <8,16,128,8>, cpg 4, 256 slots (> 128: folded first)
    (18, 1, 64, 32, 32, 256, 3, False),     # halo <8,16,128,4>, cpg 8
    (6, 2, 320, 32, 32, 320, 1, True),      # dma <64,160,3> 1x1 (proj_out + residual), cpg 10
    (7, 2, 64, 16, 16, 640, 3, True),       # dma <128,160,3>
    (9, 2, 640, 16, 16, 640, 1, True),      # dma <32,160,4>
    (30, 1, 64, 32, 32, 320, 3, False),     # four-wave <64,160,2>
    (28, 1, 64, 64, 64, 512, 3, False),     # four-wave <128,128,2>, cpg 16 (strips of 64)
]


@pytest.mark.parametrize("case", CASES)
def test_conv_emits_groupnorm_partials(ops, case):
    from stablediffusioneo_amd import _lib
    lib = _lib.load()
    tile, n, cin, h, w, cout, k, with_res = case
    x = h16(randn((n, h, w, cin), 500 + tile)).to(DEV)
    wt = h16(randn((cout, k, k, cin), 501) * (1.0 / (cin * k * k)) ** 0.5).to(DEV)
    bias = (0.5 * randn((cout,), 502) + 0.3).to(DEV)           # a non-zero mean exercises the E[x^2] - E[x]^2 form
    res = h16(randn((n, h, w, cout), 503)).to(DEV) if with_res else None
    gamma = (1.0 + 0.2 * randn((cout,), 504)).to(DEV)
    beta = (0.1 * randn((cout,), 505)).to(DEV)
    try:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(1))
        got = ops.conv2d_gn(x, wt, gamma, beta, bias=bias, res=res, eps=1e-5, swish=True)
        assert got is not None, f"tile {tile}: the plan refused to emit partials for {case}"
        y, yn, slots = got
        y_plain = ops.conv2d_nhwc(x, wt, bias=bias, res=res)
        got2 = ops.conv2d_gn(x, wt, gamma, beta, bias=bias, res=res, eps=1e-5, swish=True)
    finally:
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
    assert torch.equal(y, y_plain), "emitting the partials changed the conv's output"
    assert torch.equal(yn, got2[1]), "not deterministic"
    ref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    err = (yn.float() - ref).abs()
    tol = 2e-3 * ref.abs() + 3e-3
    assert bool((err <= tol).all()), f"{case}: slots {slots}, max err {float(err.max()):.3e}"
    two_pass = ops.groupnorm_nhwc(y, gamma, beta, 32, 1e-5, True)
    assert float((yn.float() - two_pass.float()).abs().max()) <= 4e-3


def test_plans_that_cannot_emit_are_refused(ops):
    """split-K, a strip that cuts a group (64-wide strips, groups of 10 channels) and the register-staged fallback kernel return no
    partials (the networks then keep the statistics pass)."""
    from stablediffusioneo_amd import _lib
    lib = _lib.load()
    x = h16(randn((1, 32, 32, 128), 510)).to(DEV)           # two 64-channel slices: split-K 2 is a real split for the halo kernel
    wt = h16(randn((320, 3, 3, 128), 511) * 0.03).to(DEV)
    g, b = torch.ones(320, device=DEV), torch.zeros(320, device=DEV)
    try:
        for tile, sk in [(13, 2), (0, 1), (17, 1)]:          # halo split-K; dma <128,128,3> (TN 64, cpg 10); halo BN 64
            lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
            assert ops.conv2d_gn(x, wt, g, b) is None, (tile, sk)
    finally:
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
    x8 = h16(randn((1, 32, 32, 8), 512)).to(DEV)             # Cin % 64 != 0: the fallback kernel
    w8 = h16(randn((320, 3, 3, 8), 513) * 0.1).to(DEV)
    assert ops.conv2d_gn(x8, w8, g, b) is None
