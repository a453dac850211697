"""Split-K reduce fused into the following single-launch GroupNorm (csrc/net.hip: fuse_reduce_groupnorm, csrc/norm.hip: GnReduce).
The GroupNorm kernel sums the conv's fp32 partial slabs with the arithmetic of splitk_reduce_kernel (same order, same single
rounding), so `apply_model` must return the SAME BITS with the fusion on and off, at fp16 and with fp8 weights (whose per-channel
scale is applied in the reduce), and the fused pairs must actually exist at the benchmarked size."""
import pytest
import torch

from tests.common import make_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [16, 8])
def test_reduce_groupnorm_fusion_is_bit_identical_sd15(bits):
    from stablediffusioneo_amd import spec as S
    from stablediffusioneo_amd.runtime import SdeoRuntime
    rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, weight_bits=bits)
    rt.load_synthetic(0)
    scales = [0.825 ** (12 - i) for i in range(13)]
    for (n, h, w) in [(2, 64, 64), (2, 32, 48), (1, 16, 16)]:
        rt.configure(n, h, w)
        fused = rt.lib.sdeo_debug_reduce_gn_count(rt.handle)
        x, ctx, hint = make_inputs(n, h, w, ctx_dim=S.UNET_SD15.context_dim)
        tt = torch.tensor([801, 1][:n], dtype=torch.long)
        out = []
        for on in (1, 0, 1):
            assert rt.lib.sdeo_debug_set_reduce_gn(rt.handle, on) == 0
            out.append(rt.apply_model(x, hint, tt, ctx, scales=scales).clone())
            ctrl = [c.clone() for c in rt.controlnet(x, hint, tt, ctx)]
            out.append(torch.cat([c.flatten() for c in ctrl]))
        assert rt.lib.sdeo_debug_set_reduce_gn(rt.handle, 0) == 0          # the default (measured neutral on the step)
        print(f"[fusion] sd15 n{n} {h}x{w} fp{bits}: {fused} [split-K conv, GroupNorm] pairs fused per pass")
        assert fused >= 10, fused                       # the ResBlocks of the 16x16 / 8x8 levels
        assert torch.isfinite(out[0]).all() and float(out[0].abs().max()) > 0
        assert torch.equal(out[0], out[2]) and torch.equal(out[0], out[4]), float((out[0] - out[2]).abs().max())
        assert torch.equal(out[1], out[3])
