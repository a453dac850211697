"""Pair launches (csrc/net.hip: merge_pairs): the ControlNet and the UNet encoder + middle block zipped into one program whose
same-shaped ops share a launch.  The arithmetic of each problem is untouched, so `apply_model` must return the SAME BITS as the
two-program schedule — on the tiny config, on the full SD-1.5 config (where split-K plans, halo kernels, the LayerNorm-folded
GEMMs, attention d = 40 / 80 / 160 and both GroupNorm variants all occur), with scaled controls, and with fp8 weights.  The
paired program is an option (SDEO_PAIR=1): it halves the launches of that phase but measured no faster than two streams."""
import ctypes as C

import pytest
import torch

from tests.common import make_inputs

pytestmark = pytest.mark.gpu


def _runtime(unet_cfg, vae_cfg, bits):
    from stablediffusioneo_amd.runtime import SdeoRuntime
    rt = SdeoRuntime(unet_cfg, vae_cfg, weight_bits=bits)
    rt.load_synthetic(0)
    return rt


def _both(rt, n, h, w, ctx_dim, scales):
    rt.configure(n, h, w)
    shared, single = C.c_int(0), C.c_int(0)
    assert rt.lib.sdeo_debug_pair_counts(rt.handle, C.byref(shared), C.byref(single)) == 0
    x, ctx, hint = make_inputs(n, h, w, ctx_dim=ctx_dim)
    tt = torch.tensor([801, 1][:n], dtype=torch.long)
    out = []
    for on in (1, 0, 1):
        assert rt.lib.sdeo_debug_set_pair(rt.handle, on) == 0
        out.append(rt.apply_model(x, hint, tt, ctx, scales=scales).clone())
    assert rt.lib.sdeo_debug_set_pair(rt.handle, 0) == 0
    return out, shared.value, single.value


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (1, 8, 24)])
def test_pair_program_is_bit_identical_tiny(n, h, w):
    from stablediffusioneo_amd import spec as S
    rt = _runtime(S.UNET_TINY, S.VAE_TINY, 16)
    (a, b, c), shared, single = _both(rt, n, h, w, S.UNET_TINY.context_dim, [1.0] * 13)
    assert shared > single, (shared, single)           # nearly everything pairs: the extras are the 13 zero convs and the input staging
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("bits", [16, 8])
def test_pair_program_is_bit_identical_sd15(bits):
    from stablediffusioneo_amd import spec as S
    rt = _runtime(S.UNET_SD15, S.VAE_SD15, bits)
    scales = [0.825 ** (12 - i) for i in range(13)]
    for (n, h, w) in [(2, 64, 64), (2, 32, 48), (1, 8, 8)]:
        (a, b, c), shared, single = _both(rt, n, h, w, S.UNET_SD15.context_dim, scales)
        print(f"[pair] sd15 n{n} {h}x{w} fp{bits}: {shared} shared launches, {single} single")
        assert shared > 2 * single, (shared, single)     # singles: 13 zero convs, input staging, row_stats after split-K plans
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(a, b) and torch.equal(a, c), (n, h, w, float((a - b).abs().max()))
