"""Pin the oracle (oracle/sd_oracle.py) and the parameter inventory (spec.py) against the golden
fixtures produced by the reference's own modules (tests/golden/make_golden.py).

Tolerance = the reference's own torch<->ONNX convention, rtol 1e-3 / atol 1e-5
(`export_onnx_all.py:76`); the sampler and schedule arrays are compared tighter."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sd_oracle as O
from stablediffusioneo_amd import spec as S
from tests.common import GOLDEN, make_hint, make_inputs, randn

RTOL, ATOL = 1e-3, 1e-5


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---------------------------------------------------------------- parameter inventory

def test_manifest_full_sd15():
    man = json.load(open(os.path.join(GOLDEN, "manifest_sd15.json")))
    su = {k: list(v) for k, v in S.param_spec_unet().items()}
    sc = {k: list(v) for k, v in S.param_spec_controlnet().items()}
    sv = {k[len("decoder."):]: list(v) for k, v in S.param_spec_vae().items() if k.startswith("decoder.")}
    assert su == man["unet"] and len(su) == 686
    assert sc == man["controlnet"] and len(sc) == 340
    assert sv == man["decoder"]
    # 859.5 M / 361.3 M / 49.5 M parameters
    assert round(S.count_params(su) / 1e6, 1) == 859.5
    assert round(S.count_params(sc) / 1e6, 1) == 361.3
    assert round(S.count_params(sv) / 1e6, 1) == 49.5


def test_synth_weights_deterministic():
    a = S.synth_tensor("model.diffusion_model.out.2.weight", (4, 320, 3, 3), 0)
    b = S.synth_tensor("model.diffusion_model.out.2.weight", (4, 320, 3, 3), 0)
    assert torch.equal(a, b) and a.abs().max() > 0
    assert not torch.equal(a, S.synth_tensor("model.diffusion_model.out.2.weight", (4, 320, 3, 3), 1))


# ---------------------------------------------------------------- blocks

@pytest.fixture(scope="module")
def blocks():
    return np.load(os.path.join(GOLDEN, "blocks.npz"))


def _sd(spec, prefix, strip):
    sd = S.synth_state_dict(spec, 1, prefix)
    return {strip + k: v for k, v in sd.items()}


@pytest.mark.parametrize("tag,cin,cout", [("res_id", 64, 64), ("res_skip", 96, 64)])
def test_resblock(blocks, tag, cin, cout):
    spec = {}
    S._res(spec, S.Block("res", "rb", cin, cout), 256)
    spec = {k[len("rb."):]: v for k, v in spec.items()}
    sd = {"rb." + k: v for k, v in S.synth_state_dict(spec, 1, tag + ".").items()}
    y = O.res_block(sd, "rb", randn((2, cin, 12, 20), 12), randn((2, 256), 11))
    close(y, blocks[tag])


@pytest.mark.parametrize("tag,c,heads,ctxd,hw", [("st_c64", 64, 8, 96, (8, 8)), ("st_c320", 320, 8, 768, (16, 16)),
                                                 ("st_c128", 128, 4, 96, (4, 12))])
def test_spatial_transformer(blocks, tag, c, heads, ctxd, hw):
    spec = {}
    S._attn(spec, S.Block("attn", "st", c, c, heads), ctxd)
    spec = {k[len("st."):]: v for k, v in spec.items()}
    sd = {"st." + k: v for k, v in S.synth_state_dict(spec, 1, tag + ".").items()}
    y = O.spatial_transformer(sd, "st", randn((2, c, *hw), 13), randn((2, 77, ctxd), 14), heads)
    close(y, blocks[tag])


@pytest.mark.parametrize("tag,c,heads,ctxd,n", [("ca_self_d40", 320, 8, None, 64), ("ca_cross_d40", 320, 8, 768, 256),
                                                ("ca_self_d160", 1280, 8, None, 64), ("ca_cross_d80", 640, 8, 768, 96)])
def test_cross_attention(blocks, tag, c, heads, ctxd, n):
    spec = {}
    kd = c if ctxd is None else ctxd
    S._lin(spec, "to_q", c, c, False); S._lin(spec, "to_k", kd, c, False); S._lin(spec, "to_v", kd, c, False)
    S._lin(spec, "to_out.0", c, c)
    sd = {"ca." + k: v for k, v in S.synth_state_dict(spec, 1, tag + ".").items()}
    ctx = None if ctxd is None else randn((2, 77, ctxd), 16)
    y = O.cross_attention(sd, "ca", randn((2, n, c), 15), ctx, heads)
    close(y, blocks[tag])


def test_down_up(blocks):
    spec = {}
    S._conv(spec, "op", 64, 64, 3)
    sd = {"d.0." + k: v for k, v in S.synth_state_dict(spec, 1, "down.").items()}
    x = randn((2, 64, 10, 14), 17)
    close(O._run_blocks(sd, [S.Block("down", "d.0", 64, 64)], x, None, None), blocks["down"])
    spec = {}
    S._conv(spec, "conv", 64, 64, 3)
    sd = {"u.0." + k: v for k, v in S.synth_state_dict(spec, 1, "up.").items()}
    close(O._run_blocks(sd, [S.Block("up", "u.0", 64, 64)], x, None, None), blocks["up"])


def test_vae_attn(blocks):
    spec = {}
    S._norm(spec, "norm", 128)
    for n in ("q", "k", "v", "proj_out"):
        S._conv(spec, n, 128, 128, 1)
    sd = {"a." + k: v for k, v in S.synth_state_dict(spec, 1, "vattn.").items()}
    close(O._vae_attn(sd, "a", randn((1, 128, 8, 12), 18)), blocks["vae_attn"])


def test_hint_block(blocks):
    convs = S.hint_block_convs(S.UNET_TINY)
    hspec = {}
    for name, ci, co, _ in convs:
        S._conv(hspec, name[len("input_hint_block."):], ci, co, 3)
    sd = {"input_hint_block." + k: v for k, v in S.synth_state_dict(hspec, 1, "hint.").items()}
    close(O.hint_block(sd, make_hint(2, 64, 96), convs), blocks["hint_block"])


def test_timestep_embedding(blocks):
    t = torch.tensor([1, 51, 501, 951, 981], dtype=torch.long)
    close(O.timestep_embedding(t, 320), blocks["timestep_embedding_320"], rtol=1e-6, atol=1e-6)


def test_group_norm(blocks):
    x = randn((2, 96, 6, 10), 19) * 3.0 + 0.5
    sd = {"g.weight": S.synth_tensor("gn.weight.norm", (96,), 1), "g.bias": S.synth_tensor("gn.bias", (96,), 1)}
    close(O.group_norm(sd, "g", x, 1e-5), blocks["gn_eps5"])
    close(torch.nn.functional.silu(O.group_norm(sd, "g", x, 1e-6)), blocks["gn_eps6_silu"])


# ---------------------------------------------------------------- nets (reduced config)

@pytest.fixture(scope="module")
def tiny():
    ucfg, vcfg = S.UNET_TINY, S.VAE_TINY
    return dict(
        gold=np.load(os.path.join(GOLDEN, "tiny_nets.npz")),
        su=S.synth_state_dict(S.param_spec_unet(ucfg), 0, S.NS_UNET),
        sc=S.synth_state_dict(S.param_spec_controlnet(ucfg), 0, S.NS_CONTROL),
        sv=S.synth_state_dict(S.param_spec_vae(vcfg), 0, S.NS_VAE),
        uplan=S.unet_plan(ucfg), cplan=S.unet_plan(ucfg, with_decoder=False),
        hint_convs=S.hint_block_convs(ucfg), levels=S.vae_plan(vcfg)[1])


@pytest.mark.parametrize("n,h,w,t", [(2, 16, 16, [801, 1]), (1, 8, 24, [401])])
def test_tiny_nets(tiny, n, h, w, t):
    g = tiny["gold"]
    tag = f"n{n}_{h}x{w}"
    x, ctx, hint = make_inputs(n, h, w, ctx_dim=S.UNET_TINY.context_dim)
    t = torch.tensor(t, dtype=torch.long)
    with torch.no_grad():
        ctrl = O.controlnet_forward(tiny["sc"], tiny["cplan"], tiny["hint_convs"], x, hint, t, ctx)
        assert len(ctrl) == 13
        for i, c in enumerate(ctrl):
            close(c, g[f"{tag}.control{i}"])
        close(O.unet_forward(tiny["su"], tiny["uplan"], x, t, ctx, ctrl), g[f"{tag}.eps"])
        close(O.unet_forward(tiny["su"], tiny["uplan"], x, t, ctx, None), g[f"{tag}.eps_nocontrol"])
        eps = O.apply_model(tiny["su"], tiny["sc"], tiny["uplan"], tiny["cplan"], tiny["hint_convs"], x, t, ctx,
                            hint, [1.0] * 13)
        close(eps, g[f"{tag}.eps"])
        sv = {k: v for k, v in tiny["sv"].items()}
        close(O.vae_decoder(sv, tiny["levels"], x), g[f"{tag}.dec"])


# ---------------------------------------------------------------- sampler / schedule

@pytest.fixture(scope="module")
def sampler():
    return np.load(os.path.join(GOLDEN, "sampler.npz"))


def test_schedule(sampler):
    sch = O.register_schedule()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev"):
        np.testing.assert_allclose(sch[k].numpy(), sampler[k], rtol=1e-6, atol=0)
    # SURVEY.md A1/A3 known answers
    assert list(O.make_ddim_timesteps(5)) == [1, 201, 401, 601, 801]
    assert list(O.make_ddim_timesteps(20))[:2] == [1, 51] and O.make_ddim_timesteps(20)[-1] == 951
    assert O.make_ddim_timesteps(50)[-1] == 981
    assert abs(float(sch["alphas_cumprod"][1]) - 0.9983) < 1e-4
    assert abs(float(sch["alphas_cumprod"][801]) - 0.0365) < 1e-4


@pytest.mark.parametrize("S_", [5, 20, 50])
def test_ddim_trajectory(sampler, S_):
    sch = O.register_schedule()
    ts = O.make_ddim_timesteps(S_)
    np.testing.assert_array_equal(ts, sampler[f"S{S_}.timesteps"])
    sig, al, alp = O.make_ddim_sampling_parameters(sch["alphas_cumprod"].numpy(), ts, 0.0)
    np.testing.assert_allclose(al, sampler[f"S{S_}.alphas"], rtol=1e-6)
    np.testing.assert_allclose(alp, sampler[f"S{S_}.alphas_prev"], rtol=1e-6)

    def apply_model(x, t, c):
        return torch.tanh(x * c) * 0.7 + 0.1 * torch.sin(t.float() / 100.0)[:, None, None, None] * x.roll(1, -1)

    x_T = randn((2, 4, 8, 8), 2946901)
    x0, inter = O.ddim_sample(apply_model, x_T, S_, torch.full((2, 1, 1, 1), 0.9), torch.full((2, 1, 1, 1), -0.4), 9.0)
    close(x0, sampler[f"S{S_}.x0"], rtol=1e-4, atol=1e-5)
    close(torch.stack(inter["x_inter"]), sampler[f"S{S_}.x_inter"], rtol=1e-4, atol=1e-5)


def test_sigmas_eta(sampler):
    sch = O.register_schedule()
    ts = O.make_ddim_timesteps(20)
    sig, _, _ = O.make_ddim_sampling_parameters(sch["alphas_cumprod"].numpy(), ts, 0.5)
    np.testing.assert_allclose(sig, sampler["S20.sigmas_eta0.5"], rtol=1e-6)


# ---------------------------------------------------------------- the reference's own op test

def test_reference_attention_vectors():
    g = np.load(os.path.join(GOLDEN, "attention_test.npz"))
    sd = {"a." + k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd.")}
    y = O.cross_attention(sd, "a", torch.tensor(g["x"]), torch.tensor(g["context"]), 8)
    # the reference's tolerance for this test: atol 1e-6 (`test_attention_onnx_torch_error.py:198`)
    np.testing.assert_allclose(y.numpy(), g["out_original"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(y.numpy(), g["out_fused_class"], atol=1e-6, rtol=0)


# ---------------------------------------------------------------- full SD-1.5 config, latent 8x8

@pytest.mark.slow
def test_full_sd15_lat8():
    path = os.path.join(GOLDEN, "sd15_lat8.npz")
    if not os.path.exists(path):
        pytest.skip("sd15_lat8.npz not generated")
    g = np.load(path)
    ucfg = S.UNET_SD15
    su = S.synth_state_dict(S.param_spec_unet(ucfg), 0, S.NS_UNET)
    sc = S.synth_state_dict(S.param_spec_controlnet(ucfg), 0, S.NS_CONTROL)
    x, ctx, hint = make_inputs(2, 8, 8)
    t = torch.tensor([801, 801], dtype=torch.long)
    with torch.no_grad():
        ctrl = O.controlnet_forward(sc, S.unet_plan(ucfg, False), S.hint_block_convs(ucfg), x, hint, t, ctx)
        # same rtol as everywhere; atol scaled by the tensor's magnitude (outputs reach |y| ~ 15 after ~60 fp32 layers
        # whose reductions run in a different thread split here than inside the reference's nn.Modules)
        for i, c in enumerate(ctrl):
            close(c, g[f"control{i}"], atol=ATOL * max(1.0, float(np.abs(g[f"control{i}"]).max())))
        close(O.unet_forward(su, S.unet_plan(ucfg), x, t, ctx, ctrl), g["eps"], atol=ATOL * max(1.0, float(np.abs(g["eps"]).max())))


# ---------------------------------------------------------------- benchmarked configurations (tests/golden/sd15_full.npz)

def _full():
    path = os.path.join(GOLDEN, "sd15_full.npz")
    if not os.path.exists(path):
        pytest.skip("sd15_full.npz not generated")
    return np.load(path)


@pytest.mark.parametrize("hh", [8, 16])
def test_sd15_vae_decoder_small_latents(hh):
    """The oracle's Decoder restatement with the SD-1.5 VAE configuration (ch 128, 512-channel mid block, d = 512 attention)
    against the reference Decoder's own output."""
    g = _full()
    v = S.VAE_SD15
    sv = S.synth_state_dict(S.param_spec_vae(v), 0, S.NS_VAE)
    with torch.no_grad():
        img = O.decode_first_stage(sv, S.vae_plan(v)[1], torch.tensor(g[f"vae{hh}.z"]), v.scale_factor)
    ref = g[f"vae{hh}.image"]
    close(img, ref, atol=ATOL * max(1.0, float(np.abs(ref).max())))


@pytest.mark.slow
def test_full_sd15_pass64():
    """BASELINE configs[1] shape (latent 64x64, fused CFG pair N=2): oracle vs the reference modules' eps and controls."""
    g = _full()
    ucfg = S.UNET_SD15
    su = S.synth_state_dict(S.param_spec_unet(ucfg), 0, S.NS_UNET)
    sc = S.synth_state_dict(S.param_spec_controlnet(ucfg), 0, S.NS_CONTROL)
    x1 = randn((1, 4, 64, 64), 2946901)
    from tests.common import make_hint
    hint1 = make_hint(1, 512, 512)
    x, hint = torch.cat([x1, x1]), torch.cat([hint1, hint1])
    ctx = torch.cat([randn((1, 77, 768), 1), randn((1, 77, 768), 2)])
    t = torch.tensor([951, 951], dtype=torch.long)
    with torch.no_grad():
        ctrl = O.controlnet_forward(sc, S.unet_plan(ucfg, False), S.hint_block_convs(ucfg), x, hint, t, ctx)
        for i, c in enumerate(ctrl):
            ref = g[f"pass64.control{i}.sub"]
            close(c[:, ::40].contiguous(), ref, atol=ATOL * max(1.0, float(g[f"pass64.control{i}.stats"][0])))
        eps = O.unet_forward(su, S.unet_plan(ucfg), x, t, ctx, ctrl)
        close(eps, g["pass64.eps"], atol=ATOL * max(1.0, float(np.abs(g["pass64.eps"]).max())))


def test_eta_noise_trajectory(sampler):
    """eta = 0.5: the oracle's sampler with the reference's own noise draws (global CPU generator, seed in make_golden.py)."""
    from tests.golden.make_golden import ETA_SEED

    def apply_model(x, t, c):
        return torch.tanh(x * c) * 0.7 + 0.1 * torch.sin(t.float() / 100.0)[:, None, None, None] * x.roll(1, -1)

    gen = torch.Generator(device="cpu").manual_seed(ETA_SEED)
    x0, inter = O.ddim_sample(apply_model, randn((2, 4, 8, 8), 2946901), 20, torch.full((2, 1, 1, 1), 0.9),
                              torch.full((2, 1, 1, 1), -0.4), 9.0, eta=0.5, noise_fn=lambda shp: torch.randn(shp, generator=gen))
    close(x0, sampler["S20_eta0.5.x0"], rtol=1e-4, atol=2e-5)
    close(torch.stack(inter["x_inter"]), sampler["S20_eta0.5.x_inter"], rtol=1e-4, atol=2e-5)
