"""ORACLE -- test infrastructure only, never a product path.

CPU restatement (PyTorch fp32, functional style over a flat state dict) of the reference's CNSD
hot path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product package `stablediffusioneo_amd` must never do so.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against fixtures under
`tests/golden/` that were produced by importing the reference's own modules
(`tests/golden/make_golden.py`, run in the build container where `/root/reference` exists).
Two wrappers have no reference file to pin against (`decode_first_stage`/`AutoencoderKL.decode`
and the `LatentDiffusion` schedule buffers: the `ldm/models/` package is absent from the
reference tree) -- those two are marked "parity unpinned" below and in DESIGN.md.

Every function cites the reference file:line it follows (paths relative to /root/reference).
The third-party arithmetic underneath (conv2d, group_norm, layer_norm, linear, softmax, gelu,
interpolate) is PyTorch ATen, exactly as in the reference (requirements.txt:8-10).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# schedule  (A1-A3)
# ----------------------------------------------------------------------------------------------

def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """`ldm/modules/diffusionmodules/util.py:21-25,43` ("linear" = linspace in sqrt space, f64)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def register_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """LatentDiffusion.register_schedule -- PARITY UNPINNED (ldm/models/diffusion/ddpm.py is absent
    from the reference tree).  Upstream semantics: alphas_cumprod = cumprod(1 - betas) in f64,
    alphas_cumprod_prev = [1, ac[:-1]], all stored as f32 buffers."""
    betas = make_beta_schedule_linear(n_timestep, linear_start, linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return dict(betas=f32(betas), alphas_cumprod=f32(ac), alphas_cumprod_prev=f32(ac_prev))


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """`util.py:46-60`, 'uniform' discretisation."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """`util.py:63-74`.  `alphacums` is a numpy/torch f32 array of length num_ddpm_timesteps."""
    alphacums = np.asarray(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


# ----------------------------------------------------------------------------------------------
# building blocks (A9-A17)
# ----------------------------------------------------------------------------------------------

def timestep_embedding(timesteps, dim, max_period=10000):
    """`util.py:154-174` (repeat_only=False)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _conv(sd: SD, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _lin(sd: SD, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def group_norm(sd: SD, name, x, eps, groups=32):
    """GroupNorm32 (`util.py:217-219`, eps 1e-5 via `:202-208`) / Normalize (`attention.py:88-89`,
    `model.py:46-47`, eps 1e-6)."""
    return F.group_norm(x.float(), groups, sd[name + ".weight"], sd[name + ".bias"], eps).type(x.dtype)


def time_embed(sd: SD, timesteps, model_channels):
    """`openaimodel.py:528-533,769-770`; `cldm/cldm.py:131-136,285-286`."""
    t = timestep_embedding(timesteps, model_channels)
    return _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", t)))


def res_block(sd: SD, p, x, emb):
    """ResBlock._forward (`openaimodel.py:255-275`), no up/down, no scale-shift norm."""
    h = _conv(sd, f"{p}.in_layers.2", F.silu(group_norm(sd, f"{p}.in_layers.0", x, 1e-5)))
    emb_out = _lin(sd, f"{p}.emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = _conv(sd, f"{p}.out_layers.3", F.silu(group_norm(sd, f"{p}.out_layers.0", h, 1e-5)))
    if f"{p}.skip_connection.weight" in sd:
        x = _conv(sd, f"{p}.skip_connection", x, padding=0)
    return x + h


def cross_attention(sd: SD, p, x, context, heads):
    """CrossAttention.forward runtime branch (`attention.py:217-250`; clean original `:271-302`):
    unfused q/k/v projections without bias, q.k^T scaled in fp32, softmax, .v, to_out with bias."""
    context = x if context is None else context
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    k = F.linear(context, sd[f"{p}.to_k.weight"])
    v = F.linear(context, sd[f"{p}.to_v.weight"])
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * (d ** -0.5)
    sim = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", sim, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(sd, f"{p}.to_out.0", out)


def feed_forward(sd: SD, p, x):
    """FeedForward with GEGLU (`attention.py:49-76`): proj -> chunk -> x * gelu(gate) (erf) -> Linear."""
    h = _lin(sd, f"{p}.net.0.proj", x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, f"{p}.net.2", a * F.gelu(gate))


def _ln(sd: SD, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def basic_transformer_block(sd: SD, p, x, context, heads):
    """BasicTransformerBlock._forward (`attention.py:381-385`)."""
    x = cross_attention(sd, f"{p}.attn1", _ln(sd, f"{p}.norm1", x), None, heads) + x
    x = cross_attention(sd, f"{p}.attn2", _ln(sd, f"{p}.norm2", x), context, heads) + x
    x = feed_forward(sd, f"{p}.ff", _ln(sd, f"{p}.norm3", x)) + x
    return x


def spatial_transformer(sd: SD, p, x, context, heads):
    """SpatialTransformer.forward (`attention.py:431-450`), use_linear=False, depth=1."""
    b, c, h, w = x.shape
    x_in = x
    x = group_norm(sd, f"{p}.norm", x, 1e-6)
    x = _conv(sd, f"{p}.proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = basic_transformer_block(sd, f"{p}.transformer_blocks.0", x, context, heads)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = _conv(sd, f"{p}.proj_out", x, padding=0)
    return x + x_in


def _run_blocks(sd: SD, blocks, h, emb, context):
    """TimestepEmbedSequential.forward (`openaimodel.py:79-87`)."""
    for b in blocks:
        if b.kind == "conv_in":
            h = _conv(sd, b.name, h)
        elif b.kind == "res":
            h = res_block(sd, b.name, h, emb)
        elif b.kind == "attn":
            h = spatial_transformer(sd, b.name, h, context, b.heads)
        elif b.kind == "down":           # Downsample (`openaimodel.py:133-159`)
            h = _conv(sd, f"{b.name}.op", h, stride=2)
        elif b.kind == "up":             # Upsample (`openaimodel.py:108-118`)
            h = _conv(sd, f"{b.name}.conv", F.interpolate(h, scale_factor=2, mode="nearest"))
        else:
            raise ValueError(b.kind)
    return h


# ----------------------------------------------------------------------------------------------
# networks (A6-A8, A18-A19)
# ----------------------------------------------------------------------------------------------

def hint_block(sd: SD, hint, hint_convs):
    """input_hint_block (`cldm/cldm.py:147-163`): 8 conv3x3 with SiLU between (none after the last)."""
    h = hint
    for i, (name, _, _, stride) in enumerate(hint_convs):
        h = _conv(sd, name, h, stride=stride)
        if i != len(hint_convs) - 1:
            h = F.silu(h)
    return h


def controlnet_forward(sd: SD, plan, hint_convs, x, hint, timesteps, context) -> List[torch.Tensor]:
    """ControlNet.forward (`cldm/cldm.py:284-305`) -> 13 control tensors."""
    emb = time_embed(sd, timesteps, plan.cfg.model_channels)
    guided_hint = hint_block(sd, hint, hint_convs)
    outs = []
    h = x
    for i, blocks in enumerate(plan.input_blocks):
        h = _run_blocks(sd, blocks, h, emb, context)
        if guided_hint is not None:
            h = h + guided_hint
            guided_hint = None
        outs.append(_conv(sd, f"zero_convs.{i}.0", h, padding=0))
    h = _run_blocks(sd, plan.middle_block, h, emb, context)
    outs.append(_conv(sd, "middle_block_out.0", h, padding=0))
    return outs


def unet_forward(sd: SD, plan, x, timesteps, context, control: Optional[Sequence[torch.Tensor]],
                 only_mid_control=False):
    """ControlledUnetModel.forward (`cldm/cldm.py:22-45`).  `control` is NOT consumed here."""
    control = None if control is None else list(control)
    emb = time_embed(sd, timesteps, plan.cfg.model_channels)
    hs = []
    h = x
    for blocks in plan.input_blocks:
        h = _run_blocks(sd, blocks, h, emb, context)
        hs.append(h)
    h = _run_blocks(sd, plan.middle_block, h, emb, context)
    if control is not None:
        h = h + control.pop()
    for blocks in plan.output_blocks:
        if only_mid_control or control is None:
            h = torch.cat([h, hs.pop()], dim=1)
        else:
            h = torch.cat([h, hs.pop() + control.pop()], dim=1)
        h = _run_blocks(sd, blocks, h, emb, context)
    h = F.silu(group_norm(sd, "out.0", h, 1e-5))
    return _conv(sd, "out.2", h)


def apply_model(sd_unet: SD, sd_cn: SD, uplan, cplan, hint_convs, x, t, context, hint, control_scales,
                only_mid_control=False):
    """ControlLDM.apply_model (`cldm/cldm.py:328-341`); `hint=None` is the c_concat=None branch."""
    if hint is None:
        return unet_forward(sd_unet, uplan, x, t, context, None, only_mid_control)
    control = controlnet_forward(sd_cn, cplan, hint_convs, x, hint, t, context)
    control = [c * s for c, s in zip(control, control_scales)]
    return unet_forward(sd_unet, uplan, x, t, context, control, only_mid_control)


def _vae_res(sd: SD, p, x):
    """ResnetBlock.forward with temb=None (`model.py:129-149`)."""
    h = _conv(sd, f"{p}.conv1", F.silu(group_norm(sd, f"{p}.norm1", x, 1e-6)))
    h = _conv(sd, f"{p}.conv2", F.silu(group_norm(sd, f"{p}.norm2", h, 1e-6)))
    if f"{p}.nin_shortcut.weight" in sd:
        x = _conv(sd, f"{p}.nin_shortcut", x, padding=0)
    return x + h


def _vae_attn(sd: SD, p, x):
    """AttnBlock.forward (`model.py:179-203`): single head, scale C^-0.5, softmax over keys."""
    h_ = group_norm(sd, f"{p}.norm", x, 1e-6)
    q = _conv(sd, f"{p}.q", h_, padding=0)
    k = _conv(sd, f"{p}.k", h_, padding=0)
    v = _conv(sd, f"{p}.v", h_, padding=0)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(sd, f"{p}.proj_out", h_, padding=0)


def vae_decoder(sd: SD, levels, z, prefix="decoder"):
    """Decoder.forward (`model.py:619-652`), attn_resolutions=[] so only mid.attn_1."""
    d = prefix
    h = _conv(sd, f"{d}.conv_in", z)
    h = _vae_res(sd, f"{d}.mid.block_1", h)
    h = _vae_attn(sd, f"{d}.mid.attn_1", h)
    h = _vae_res(sd, f"{d}.mid.block_2", h)
    for i_level, blocks, has_up in levels:
        for j in range(len(blocks)):
            h = _vae_res(sd, f"{d}.up.{i_level}.block.{j}", h)
        if has_up:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{d}.up.{i_level}.upsample.conv", h)
    h = F.silu(group_norm(sd, f"{d}.norm_out", h, 1e-6))
    return _conv(sd, f"{d}.conv_out", h)


def decode_first_stage(sd: SD, levels, z, scale_factor=0.18215):
    """PARITY UNPINNED wrapper (LatentDiffusion.decode_first_stage / AutoencoderKL.decode are
    absent from the reference tree; `canny2image_torch.py:63-67` documents the semantics):
    z / scale_factor -> post_quant_conv (1x1) -> Decoder."""
    z = z / scale_factor
    z = _conv(sd, "post_quant_conv", z, padding=0)
    return vae_decoder(sd, levels, z)


def postprocess_uint8(x):
    """`canny2image_torch.py:68`: NCHW [-1,1] -> NHWC uint8."""
    y = (x.permute(0, 2, 3, 1) * 127.5 + 127.5).cpu().numpy().clip(0, 255).astype(np.uint8)
    return y


# ----------------------------------------------------------------------------------------------
# sampler (A4-A5)
# ----------------------------------------------------------------------------------------------

def ddim_step(x, e_t, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """p_sample_ddim tail (`cldm/ddim_hacked.py:208-231`), eps-parameterisation."""
    pred_x0 = (x - sqrt_one_minus_at * e_t) / math.sqrt(a_t)
    dir_xt = math.sqrt(1.0 - a_prev - sigma_t ** 2) * e_t
    x_prev = math.sqrt(a_prev) * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma_t * noise
    return x_prev, pred_x0


def ddim_sample(apply_fn, x_T, S, cond, uncond, scale, eta=0.0, schedule=None, noise_fn=None):
    """DDIMSampler.sample / ddim_sampling / p_sample_ddim (`cldm/ddim_hacked.py:54-231`) for the
    path canny2image drives: uniform timesteps, eps-parameterisation, no mask, CFG as two passes
    `e = u + s*(c-u)` (`:190-192`).  `apply_fn(x, t, cond)` -> eps.  Returns (x_0, intermediates)."""
    schedule = schedule or register_schedule()
    ts = make_ddim_timesteps(S, schedule["alphas_cumprod"].shape[0])
    ac = schedule["alphas_cumprod"].numpy()
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, eta)
    sqrt_1m = np.sqrt(1.0 - alphas)
    img = x_T
    b = img.shape[0]
    inter = {"x_inter": [img], "pred_x0": [img]}
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if uncond is None or scale == 1.0:
            e_t = apply_fn(img, t, cond)
        else:
            m_t = apply_fn(img, t, cond)
            m_u = apply_fn(img, t, uncond)
            e_t = m_u + scale * (m_t - m_u)
        noise = None
        if noise_fn is not None:
            noise = noise_fn(img.shape)
        img, pred_x0 = ddim_step(img, e_t, float(alphas[index]), float(alphas_prev[index]),
                                 float(sigmas[index]), float(sqrt_1m[index]), noise)
        inter["x_inter"].append(img)
        inter["pred_x0"].append(pred_x0)
    return img, inter


def ddim_encode(apply_fn, x0, S, t_enc, cond):
    """DDIM inversion without guidance, `cldm/ddim_hacked.py:233-279` (DDIM timesteps, `use_original_steps=False`):
    x_next = sqrt(a_next / a) x + sqrt(a_next) (sqrt(1/a_next - 1) - sqrt(1/a - 1)) eps(x, t_i)  for i < t_enc, with
    a_next = ddim_alphas[i], a = ddim_alphas_prev[i]."""
    ac = register_schedule()["alphas_cumprod"]
    ts = make_ddim_timesteps(S)
    _, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, 0.0)
    x = x0.clone()
    for i in range(t_enc):
        t = torch.full((x0.shape[0],), int(ts[i]), dtype=torch.long)
        e = apply_fn(x, t, cond)
        a_next, a = float(alphas[i]), float(alphas_prev[i])
        x = (a_next / a) ** 0.5 * x + a_next ** 0.5 * ((1 / a_next - 1) ** 0.5 - (1 / a - 1) ** 0.5) * e
    return x
