"""Checkpoint loaders (`cldm/model.py:12-28` mirror): a synthetic state dict written to `.safetensors` and to a
weights-only `.pth` (with and without the `state_dict` wrapper key the reference unwraps, `cldm/model.py:8-9`) is read back
through `load_state_dict` / `create_model(...).load_state_dict` and must give BIT-IDENTICAL network outputs to the in-memory
path.  Files are written on the box; both loaders execute nothing from the file (safetensors / weights_only=True)."""
import os

import pytest
import torch

from tests.common import make_inputs


def synthetic_checkpoint():
    from stablediffusioneo_amd import spec as S
    sd = {}
    for ns, spec in ((S.NS_UNET, S.param_spec_unet(S.UNET_TINY)), (S.NS_CONTROL, S.param_spec_controlnet(S.UNET_TINY)),
                     (S.NS_VAE, S.param_spec_vae(S.VAE_TINY))):
        for k, shp in spec.items():
            sd[ns + k] = S.synth_tensor(ns + k, shp, 0)
    # tensors the hot path must ignore (the real checkpoint carries CLIP, the VAE encoder, EMA / schedule buffers ...)
    sd["first_stage_model.encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3)
    sd["betas"] = torch.zeros(1000)
    return sd


def test_load_state_dict_reads_both_formats(tmp_path):
    """pure file I/O: runs without a GPU too"""
    import safetensors.torch
    from stablediffusioneo_amd.cldm.model import load_state_dict
    sd = synthetic_checkpoint()
    p1, p2, p3 = str(tmp_path / "ckpt.safetensors"), str(tmp_path / "ckpt.pth"), str(tmp_path / "wrapped.ckpt")
    safetensors.torch.save_file(sd, p1)
    torch.save(sd, p2)
    torch.save({"state_dict": sd, "global_step": 7}, p3)
    for p in (p1, p2, p3):
        got = load_state_dict(p)
        assert set(got) == set(sd)
        assert all(torch.equal(got[k], sd[k]) for k in sd)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["safetensors", "pth", "wrapped"])
def test_checkpoint_file_roundtrip_is_bit_identical(tmp_path, fmt):
    import safetensors.torch
    from stablediffusioneo_amd import spec as S
    from stablediffusioneo_amd.cldm.model import create_model, load_state_dict
    sd = synthetic_checkpoint()
    path = str(tmp_path / {"safetensors": "control_sd15_canny.safetensors", "pth": "control_sd15_canny.pth", "wrapped": "x.ckpt"}[fmt])
    if fmt == "safetensors":
        safetensors.torch.save_file(sd, path)
    elif fmt == "pth":
        torch.save(sd, path)
    else:
        torch.save({"state_dict": sd}, path)
    model = create_model("tiny")
    model.load_state_dict(load_state_dict(path, location="cuda"))          # `canny2image_torch.py:23` call shape
    ref = create_model("tiny")
    ref.rt.load_synthetic(0)
    n, h, w = 2, 8, 16
    x, ctx, hint = make_inputs(n, h, w, ctx_dim=S.UNET_TINY.context_dim)
    t = torch.tensor([801, 1], dtype=torch.long)
    a = model.rt.configure(n, h, w).apply_model(x, hint, t, ctx, scales=[1.0] * 13).clone()
    b = ref.rt.configure(n, h, w).apply_model(x, hint, t, ctx, scales=[1.0] * 13).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.equal(model.rt.vae_decode(x * 0.18215), ref.rt.vae_decode(x * 0.18215))
    # a missing tensor is reported by name; an unexpected one is an error only when strict
    bad = dict(sd)
    del bad[S.NS_UNET + "out.2.weight"]
    with pytest.raises(Exception, match="out.2.weight"):
        create_model("tiny").load_state_dict(bad)
    with pytest.raises(Exception, match="unexpected tensor"):
        create_model("tiny").rt.load_state_dict({"model.diffusion_model.nope": torch.zeros(1)}, strict=True)
