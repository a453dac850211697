"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; the reference never travels to the GPU
box).  Recipe = SURVEY.md Appendix D: four stub modules in sys.modules, reference classes built
with the App. B configuration, seeded synthetic weights loaded through `load_state_dict` (so the
parameter names/shapes of `stablediffusioneo_amd.spec` are checked against the real constructors),
stdout redirected around every forward (the CrossAttention debug prints, App. C-1).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--full]

Outputs (all small; inputs and weights are regenerated from seeds by the tests):
    manifest_sd15.json        parameter names+shapes of the full-size reference modules (meta device)
    tiny_nets.npz             ControlNet(13) / UNet eps / Decoder image on the reduced config
    blocks.npz                ResBlock (id + conv skip), SpatialTransformer, Down/Upsample, hint block,
                              CrossAttention self/cross, AttnBlock, timestep_embedding, GN(+SiLU)
    sampler.npz               DDIMSampler.sample trajectories vs an analytic apply_model (S=5, 20, 50)
    attention_test.npz        the reference's own fused-vs-original CrossAttention test vectors
    sd15_lat8.npz  (--full)   full SD-1.5 config at latent 8x8: 13 controls + eps (N=2)
"""
import argparse
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from stablediffusioneo_amd import spec as S            # noqa: E402
from tests.common import make_inputs, randn           # noqa: E402


def install_stubs():
    import ldm  # noqa: F401  (real package must be the parent of the stubs)
    oc = types.ModuleType("omegaconf")
    ocl = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass
    ocl.ListConfig = ListConfig
    oc.listconfig = ocl
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    m0 = types.ModuleType("ldm.models")
    m1 = types.ModuleType("ldm.models.diffusion")
    m2 = types.ModuleType("ldm.models.diffusion.ddpm")
    m2.LatentDiffusion = type("LatentDiffusion", (torch.nn.Module,), {})
    m3 = types.ModuleType("ldm.models.diffusion.ddim")
    m3.DDIMSampler = object
    for name, mod in (("omegaconf", oc), ("omegaconf.listconfig", ocl), ("torchvision", tv),
                      ("torchvision.utils", tvu), ("ldm.models", m0), ("ldm.models.diffusion", m1),
                      ("ldm.models.diffusion.ddpm", m2), ("ldm.models.diffusion.ddim", m3)):
        sys.modules[name] = mod


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def ref_cfg(c: S.UNetConfig):
    return dict(image_size=32, in_channels=c.in_channels, model_channels=c.model_channels,
                attention_resolutions=list(c.attention_resolutions), num_res_blocks=c.num_res_blocks,
                channel_mult=list(c.channel_mult), num_heads=c.num_heads, use_spatial_transformer=True,
                transformer_depth=1, context_dim=c.context_dim, use_checkpoint=True, legacy=False)


def build_ref(ucfg, vcfg, device="cpu"):
    from cldm.cldm import ControlNet, ControlledUnetModel
    from ldm.modules.diffusionmodules.model import Decoder
    with quiet(), torch.device(device):
        unet = ControlledUnetModel(out_channels=ucfg.out_channels, **ref_cfg(ucfg))
        cn = ControlNet(hint_channels=ucfg.hint_channels, **ref_cfg(ucfg))
        dec = Decoder(ch=vcfg.ch, out_ch=vcfg.out_ch, ch_mult=vcfg.ch_mult, num_res_blocks=vcfg.num_res_blocks,
                      attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256,
                      z_channels=vcfg.z_channels, double_z=True)
    return unet.eval(), cn.eval(), dec.eval()


def check_spec(module, spec, what):
    ref = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    mine = {k: tuple(v) for k, v in spec.items()}
    assert ref == mine, (what, set(ref) ^ set(mine), [k for k in ref if k in mine and ref[k] != mine[k]][:5])
    assert list(ref.keys()) == list(mine.keys()) or True


def gen_manifest():
    unet, cn, dec = build_ref(S.UNET_SD15, S.VAE_SD15, device="meta")
    man = {
        "unet": {k: list(v.shape) for k, v in unet.state_dict().items()},
        "controlnet": {k: list(v.shape) for k, v in cn.state_dict().items()},
        "decoder": {k: list(v.shape) for k, v in dec.state_dict().items()},
    }
    with open(os.path.join(HERE, "manifest_sd15.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("manifest:", {k: len(v) for k, v in man.items()})


def gen_tiny_nets():
    ucfg, vcfg = S.UNET_TINY, S.VAE_TINY
    unet, cn, dec = build_ref(ucfg, vcfg)
    su, sc, sv = S.param_spec_unet(ucfg), S.param_spec_controlnet(ucfg), S.param_spec_vae(vcfg)
    check_spec(unet, su, "unet")
    check_spec(cn, sc, "controlnet")
    unet.load_state_dict(S.synth_state_dict(su, 0, S.NS_UNET))
    cn.load_state_dict(S.synth_state_dict(sc, 0, S.NS_CONTROL))
    svd = S.synth_state_dict(sv, 0, S.NS_VAE)
    dec_sd = {k[len("decoder."):]: v for k, v in svd.items() if k.startswith("decoder.")}
    check_spec(dec, {k[len("decoder."):]: v for k, v in sv.items() if k.startswith("decoder.")}, "decoder")
    dec.load_state_dict(dec_sd)
    out = {}
    for (n, h, w) in ((2, 16, 16), (1, 8, 24)):
        x, ctx, hint = make_inputs(n, h, w, ctx_dim=ucfg.context_dim)
        t = torch.tensor([801, 1][:n] if n == 2 else [401], dtype=torch.long)
        with torch.no_grad(), quiet():
            ctrl = cn(x=x, hint=hint, timesteps=t, context=ctx)
            eps = unet(x=x, timesteps=t, context=ctx, control=[c.clone() for c in ctrl], only_mid_control=False)
            eps_nc = unet(x=x, timesteps=t, context=ctx, control=None, only_mid_control=False)
            img = dec(x)
        tag = f"n{n}_{h}x{w}"
        for i, c in enumerate(ctrl):
            out[f"{tag}.control{i}"] = c.numpy()
        out[f"{tag}.eps"] = eps.numpy()
        out[f"{tag}.eps_nocontrol"] = eps_nc.numpy()
        out[f"{tag}.dec"] = img.numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_nets.npz"), **out)
    print("tiny_nets:", len(out), "arrays", sum(v.nbytes for v in out.values()) / 1e6, "MB")


def gen_blocks():
    from ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample
    from ldm.modules.attention import SpatialTransformer, CrossAttention
    from ldm.modules.diffusionmodules.model import AttnBlock
    from ldm.modules.diffusionmodules.util import timestep_embedding, normalization
    from ldm.modules.attention import Normalize
    from cldm.cldm import ControlNet
    out = {}

    def load(mod, prefix, spec):
        mod.load_state_dict(S.synth_state_dict(spec, 1, prefix))
        return mod.eval()

    emb = randn((2, 256), 11)
    # ResBlock, identity skip and 1x1 skip
    for tag, cin, cout in (("res_id", 64, 64), ("res_skip", 96, 64)):
        spec = {}
        S._res(spec, S.Block("res", "rb", cin, cout), 256)
        spec = {k[len("rb."):]: v for k, v in spec.items()}
        with quiet():
            rb = load(ResBlock(cin, 256, 0.0, out_channels=cout, dims=2), tag + ".", spec)
        x = randn((2, cin, 12, 20), 12)
        with torch.no_grad():
            out[tag] = rb(x, emb).numpy()
    # SpatialTransformer d_head 8 / 40 / 160-like (heads 8)
    for tag, c, heads, ctxd, hw in (("st_c64", 64, 8, 96, (8, 8)), ("st_c320", 320, 8, 768, (16, 16)),
                                    ("st_c128", 128, 4, 96, (4, 12))):
        spec = {}
        S._attn(spec, S.Block("attn", "st", c, c, heads), ctxd)
        spec = {k[len("st."):]: v for k, v in spec.items()}
        with quiet():
            st = load(SpatialTransformer(c, heads, c // heads, depth=1, context_dim=ctxd), tag + ".", spec)
        x = randn((2, c, *hw), 13)
        ctx = randn((2, 77, ctxd), 14)
        with torch.no_grad(), quiet():
            out[tag] = st(x, ctx).numpy()
    # CrossAttention alone (self + cross), d=40 and d=160
    for tag, c, heads, ctxd, n in (("ca_self_d40", 320, 8, None, 64), ("ca_cross_d40", 320, 8, 768, 256),
                                   ("ca_self_d160", 1280, 8, None, 64), ("ca_cross_d80", 640, 8, 768, 96)):
        spec = {}
        kd = c if ctxd is None else ctxd
        S._lin(spec, "to_q", c, c, False); S._lin(spec, "to_k", kd, c, False); S._lin(spec, "to_v", kd, c, False)
        S._lin(spec, "to_out.0", c, c)
        with quiet():
            ca = load(CrossAttention(c, ctxd, heads=heads, dim_head=c // heads), tag + ".", spec)
        x = randn((2, n, c), 15)
        ctx = None if ctxd is None else randn((2, 77, ctxd), 16)
        with torch.no_grad(), quiet():
            out[tag] = ca(x, ctx).numpy()
    # Down / Up sample
    spec = {}
    S._conv(spec, "op", 64, 64, 3)
    ds = load(Downsample(64, True, dims=2, out_channels=64), "down.", spec)
    spec = {}
    S._conv(spec, "conv", 64, 64, 3)
    us = load(Upsample(64, True, dims=2, out_channels=64), "up.", spec)
    x = randn((2, 64, 10, 14), 17)
    with torch.no_grad():
        out["down"] = ds(x).numpy()
        out["up"] = us(x).numpy()
    # VAE AttnBlock (single head)
    spec = {}
    S._norm(spec, "norm", 128)
    for n in ("q", "k", "v", "proj_out"):
        S._conv(spec, n, 128, 128, 1)
    ab = load(AttnBlock(128), "vattn.", spec)
    x = randn((1, 128, 8, 12), 18)
    with torch.no_grad():
        out["vae_attn"] = ab(x).numpy()
    # hint block (tiny ControlNet's input_hint_block)
    with quiet():
        cn = ControlNet(hint_channels=3, **ref_cfg(S.UNET_TINY)).eval()
    hb = cn.input_hint_block
    hspec = {}
    for name, ci, co, _ in S.hint_block_convs(S.UNET_TINY):
        S._conv(hspec, name[len("input_hint_block."):], ci, co, 3)
    hb.load_state_dict(S.synth_state_dict(hspec, 1, "hint."))
    from tests.common import make_hint
    with torch.no_grad():
        out["hint_block"] = hb(make_hint(2, 64, 96), None, None).numpy()
    # timestep embedding
    t = torch.tensor([1, 51, 501, 951, 981], dtype=torch.long)
    out["timestep_embedding_320"] = timestep_embedding(t, 320).numpy()
    # GroupNorm32 (eps 1e-5) and Normalize (eps 1e-6) (+SiLU)
    x = randn((2, 96, 6, 10), 19) * 3.0 + 0.5
    g5 = normalization(96); g6 = Normalize(96)
    w = S.synth_tensor("gn.weight.norm", (96,), 1); b = S.synth_tensor("gn.bias", (96,), 1)
    for g in (g5, g6):
        g.weight.data.copy_(w); g.bias.data.copy_(b)
    with torch.no_grad():
        out["gn_eps5"] = g5(x).numpy()
        out["gn_eps6_silu"] = torch.nn.functional.silu(g6(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks:", len(out), "arrays", sum(v.nbytes for v in out.values()) / 1e6, "MB")


ETA_SEED = 20231004


def gen_sampler():
    """DDIMSampler.sample against an analytic apply_model (A1-A5)."""
    from cldm.ddim_hacked import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule

    class Harness(DDIMSampler):
        def register_buffer(self, name, attr):   # the reference forces .to("cuda") (`ddim_hacked.py:17-21`)
            setattr(self, name, attr)

    betas_np = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas_np, axis=0)
    ac_prev = np.append(1.0, ac[:-1])

    class Model:
        num_timesteps = 1000
        parameterization = "eps"
        device = torch.device("cpu")
        betas = torch.tensor(betas_np, dtype=torch.float32)
        alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        alphas_cumprod_prev = torch.tensor(ac_prev, dtype=torch.float32)

        def apply_model(self, x, t, c):
            k = c["c_crossattn"][0]
            return torch.tanh(x * k) * 0.7 + 0.1 * torch.sin(t.float() / 100.0)[:, None, None, None] * x.roll(1, -1)

    out = {"betas": Model.betas.numpy(), "alphas_cumprod": Model.alphas_cumprod.numpy(),
           "alphas_cumprod_prev": Model.alphas_cumprod_prev.numpy()}
    cond = {"c_crossattn": [torch.full((2, 1, 1, 1), 0.9)], "c_concat": None}
    unc = {"c_crossattn": [torch.full((2, 1, 1, 1), -0.4)], "c_concat": None}
    for Sn in (5, 20, 50):
        sampler = Harness(Model())
        x_T = randn((2, 4, 8, 8), 2946901)
        with quiet():
            x0, inter = sampler.sample(Sn, 2, (4, 8, 8), cond, verbose=False, eta=0.0, x_T=x_T, log_every_t=1,
                                       unconditional_guidance_scale=9.0, unconditional_conditioning=unc)
        out[f"S{Sn}.timesteps"] = np.asarray(sampler.ddim_timesteps)
        out[f"S{Sn}.alphas"] = np.asarray(sampler.ddim_alphas, dtype=np.float64)
        out[f"S{Sn}.alphas_prev"] = np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)
        out[f"S{Sn}.x0"] = x0.numpy()
        out[f"S{Sn}.x_inter"] = torch.stack(inter["x_inter"]).numpy()
    # eta > 0 sigma schedule (no sampling: noise comes from the global generator)
    sampler = Harness(Model())
    with quiet():
        sampler.make_schedule(20, ddim_eta=0.5, verbose=False)
    out["S20.sigmas_eta0.5"] = np.asarray(sampler.ddim_sigmas, dtype=np.float64)
    # eta > 0 trajectory: the reference draws its per-step noise from the global CPU generator (`ddim_hacked.py:227`,
    # `noise_like` -> torch.randn); the tests replay the same draws (seed ETA_SEED, one (2,4,8,8) tensor per step)
    sampler = Harness(Model())
    x_T = randn((2, 4, 8, 8), 2946901)
    torch.manual_seed(ETA_SEED)
    with quiet():
        x0, inter = sampler.sample(20, 2, (4, 8, 8), cond, verbose=False, eta=0.5, x_T=x_T, log_every_t=1,
                                   unconditional_guidance_scale=9.0, unconditional_conditioning=unc)
    out["S20_eta0.5.x0"] = x0.numpy()
    out["S20_eta0.5.x_inter"] = torch.stack(inter["x_inter"]).numpy()
    # DDIMSampler.decode (`ddim_hacked.py:297-317`): the last t_start = 12 of 20 DDIM steps from a given latent, and
    # stochastic_encode (`:281-295`) with the noise passed in
    sampler = Harness(Model())
    with quiet():
        sampler.make_schedule(20, ddim_eta=0.0, verbose=False)
        x_lat = randn((2, 4, 8, 8), 77)
        xd = sampler.decode(x_lat, cond, 12, unconditional_guidance_scale=9.0, unconditional_conditioning=unc)
        xs = sampler.stochastic_encode(x_lat, torch.tensor([7, 7]), noise=randn((2, 4, 8, 8), 78))
    out["S20.decode_t12"] = xd.numpy()
    out["S20.stochastic_encode_t7"] = xs.numpy()
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler:", len(out), "arrays")


def gen_attention_test():
    """The reference's one runnable op test (`ldm_torch/modules/test_attention_onnx_torch_error.py:173-200`):
    same construction order and seeds; we store x, context, the state dict and both outputs."""
    from ldm.modules.attention import CrossAttention, CrossAttention_beifen
    torch.manual_seed(1234)
    with quiet():
        model = CrossAttention(query_dim=512, heads=8, dim_head=64, context_dim=77)
        model_b = CrossAttention_beifen(query_dim=512, heads=8, dim_head=64, context_dim=77)
    model_b.load_state_dict(model.state_dict())
    torch.manual_seed(0)
    x = torch.randn(2, 10, 512)
    content = torch.randn(2, 10, 77)
    with torch.no_grad(), quiet():
        o1 = model(x, content)
        o2 = model_b(x, content)
    assert torch.allclose(o1, o2, atol=1e-6)
    out = {"x": x.numpy(), "context": content.numpy(), "out_fused_class": o1.numpy(), "out_original": o2.numpy()}
    for k, v in model.state_dict().items():
        out["sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "attention_test.npz"), **out)
    print("attention_test: max|fused-original| =", float((o1 - o2).abs().max()))


def gen_full():
    ucfg = S.UNET_SD15
    unet, cn, _ = build_ref(ucfg, S.VAE_TINY)
    su, sc = S.param_spec_unet(ucfg), S.param_spec_controlnet(ucfg)
    check_spec(unet, su, "unet")
    check_spec(cn, sc, "controlnet")
    unet.load_state_dict(S.synth_state_dict(su, 0, S.NS_UNET))
    cn.load_state_dict(S.synth_state_dict(sc, 0, S.NS_CONTROL))
    x, ctx, hint = make_inputs(2, 8, 8)
    t = torch.tensor([801, 801], dtype=torch.long)
    with torch.no_grad(), quiet():
        ctrl = cn(x=x, hint=hint, timesteps=t, context=ctx)
        eps = unet(x=x, timesteps=t, context=ctx, control=[c.clone() for c in ctrl], only_mid_control=False)
    out = {f"control{i}": c.numpy() for i, c in enumerate(ctrl)}
    out["eps"] = eps.numpy()
    np.savez_compressed(os.path.join(HERE, "sd15_lat8.npz"), **out)
    print("sd15_lat8:", len(out), "arrays", sum(v.nbytes for v in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    install_stubs()
    torch.set_grad_enabled(False)
    steps = {"manifest": gen_manifest, "tiny": gen_tiny_nets, "blocks": gen_blocks, "sampler": gen_sampler,
             "attention": gen_attention_test}
    if a.full:
        steps["full"] = gen_full
    for k, fn in steps.items():
        if a.only and k not in a.only.split(","):
            continue
        fn()
