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


# ---------------------------------------------------------------------------------------------------------------------------
# consumer side: conv3x3(silu(GroupNorm(x))) with the GroupNorm applied inside the halo conv kernel (KP::gn_in): the loader waves
# rewrite each staged patch in LDS, the statistics come as per-(image, slot, group) partials.  Reference: torch fp32 on the same fp16 x.
# ---------------------------------------------------------------------------------------------------------------------------
GNIN_CASES = [
    (34, 1, 2, 320, 32, 32, 320, True),     # halo <8,16,80,4,3>: 5 slices, one barrier per filter row
    (35, 1, 1, 320, 16, 32, 160, True),     # <8,16,80,8,3>
    (37, 1, 2, 128, 16, 16, 80, False),     # <8,8,80,4,3>, no SiLU (SpatialTransformer.norm semantics)
    (37, 4, 1, 1280, 16, 16, 640, True),    # split-K 4 over 20 slices: each workgroup builds the table of its own channel range
    (13, 1, 2, 320, 16, 32, 320, True),     # one barrier per tap (TPB = 1): one piece per step
    (22, 2, 1, 256, 16, 16, 128, True),     # <8,16,80,8>
    (38, 1, 1, 320, 24, 8, 320, True),      # <8,8,160,4,3>
]


@pytest.mark.parametrize("case", GNIN_CASES)
def test_conv_applies_groupnorm_of_its_input(ops, case):
    from stablediffusioneo_amd import _lib
    lib = _lib.load()
    tile, sk, n, cin, h, w, cout, swish = case
    x = h16(randn((n, h, w, cin), 600 + tile) * 1.3 + 0.4).to(DEV)
    wt = h16(randn((cout, 3, 3, cin), 601) * (1.0 / (cin * 9)) ** 0.5).to(DEV)
    bias = (0.1 * randn((cout,), 602)).to(DEV)
    gamma = (1.0 + 0.3 * randn((cin,), 603)).to(DEV)
    beta = (0.2 * randn((cin,), 604)).to(DEV)
    # partials as a producer would leave them: 3 slots per image, each over a third of the rows
    xf = x.float().reshape(n, h * w, 32, cin // 32)
    cuts = [0, (h * w) // 3, (2 * h * w) // 3, h * w]
    part = torch.stack([torch.stack([xf[:, cuts[i]:cuts[i + 1]].sum(dim=(1, 3)), (xf[:, cuts[i]:cuts[i + 1]] ** 2).sum(dim=(1, 3))], -1)
                        for i in range(3)], 1).contiguous()                  # [n][3][32][2]
    try:
        lib.sdeo_debug_force_gemm_plan(C.c_int(tile), C.c_int(sk))
        y = ops.conv3x3_gn_in(x, wt, gamma, beta, part, bias=bias, eps=1e-5, swish=swish)
        y2 = ops.conv3x3_gn_in(x, wt, gamma, beta, part, bias=bias, eps=1e-5, swish=swish)
    finally:
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
    assert y is not None, f"{case}: the plan refused"
    assert torch.equal(y, y2), "not deterministic"
    t = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)
    t = F.silu(t) if swish else t
    ref = F.conv2d(t.half().float(), wt.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    err = (y.float() - ref).abs()
    tol = 3e-3 * ref.abs() + 4e-3 * float(ref.abs().max())
    assert bool((err <= tol).all()), f"{case}: max err {float(err.max()):.3e} at scale {float(ref.abs().max()):.3g}"


def test_gn_in_is_refused_without_lds_for_the_table(ops):
    """<8,16,80,8,3> leaves 4 KB of LDS beside its ring: ten 64-channel slices of (a, b) do not fit -- the networks then keep the
    GroupNorm launch"""
    from stablediffusioneo_amd import _lib
    lib = _lib.load()
    x = h16(randn((1, 16, 32, 640), 610)).to(DEV)
    wt = h16(randn((160, 3, 3, 640), 611) * 0.01).to(DEV)
    g, b = torch.ones(640, device=DEV), torch.zeros(640, device=DEV)
    part = torch.zeros((1, 1, 32, 2), device=DEV)
    try:
        lib.sdeo_debug_force_gemm_plan(C.c_int(35), C.c_int(1))
        assert ops.conv3x3_gn_in(x, wt, g, b, part) is None
    finally:
        lib.sdeo_debug_force_gemm_plan(C.c_int(-1), C.c_int(0))
