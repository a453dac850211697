"""Architecture restatement of the CNSD hot path: configuration, parameter inventory and
seeded synthetic weights.

The reference builds its networks by running nn.Module constructors; the checkpoint key names
and tensor shapes that fall out of those constructors are the on-disk contract
(`cldm/model.py:12-21` loads a flat state dict).  This module restates that contract as data:

* `UNET_SD15` / `VAE_SD15`  -- the SD-1.5 + ControlNet-1.0 configuration.  `models/cldm_v15.yaml`
  is absent from the reference tree; the values are the ones the in-repo
  evidence requires (`export_onnx_all.py:242-256`, `Engine.py:55-91`).
* `unet_plan` / `controlnet_plan` / `vae_plan` -- block lists mirroring the constructor loops of
  `ldm/modules/diffusionmodules/openaimodel.py:544-732`, `cldm/cldm.py:131-279` and
  `ldm/modules/diffusionmodules/model.py:546-616`.
* `param_spec_*`  -- ordered `{name: shape}` for each network, same names as the reference modules.
* `synth_state_dict` -- deterministic synthetic weights from the torch CPU generator; every
  tensor is drawn from its own seed derived from (seed, name) so the oracle, the HIP path and
  the golden-fixture generator regenerate identical weights without shipping them.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch

# ----------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    hint_channels: int = 3
    model_channels: int = 320
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768
    context_len: int = 77

    @property
    def time_embed_dim(self) -> int:
        return self.model_channels * 4


@dataclass(frozen=True)
class VAEConfig:
    ch: int = 128
    out_ch: int = 3
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    embed_dim: int = 4
    scale_factor: float = 0.18215


@dataclass(frozen=True)
class ScheduleConfig:
    """LatentDiffusion schedule values (upstream SD-v1; not present in the reference tree)."""
    timesteps: int = 1000
    linear_start: float = 0.00085
    linear_end: float = 0.012
    parameterization: str = "eps"


UNET_SD15 = UNetConfig()
VAE_SD15 = VAEConfig()
SCHEDULE_SD15 = ScheduleConfig()

# reduced configuration with the same topology, for fast numeric tests
UNET_TINY = UNetConfig(model_channels=64, context_dim=96, num_heads=4)  # context_dim must differ from every query dim (attention.py:168-173 quirk)
VAE_TINY = VAEConfig(ch=32)


# ----------------------------------------------------------------------------------------
# block plans
# ----------------------------------------------------------------------------------------


@dataclass
class Block:
    """One entry of a TimestepEmbedSequential (`openaimodel.py:73-87`)."""
    kind: str                 # 'conv_in' | 'res' | 'attn' | 'down' | 'up'
    name: str                 # parameter prefix, e.g. 'input_blocks.1.0'
    cin: int
    cout: int
    heads: int = 0


@dataclass
class UNetPlan:
    cfg: UNetConfig
    input_blocks: List[List[Block]] = field(default_factory=list)
    middle_block: List[Block] = field(default_factory=list)
    output_blocks: List[List[Block]] = field(default_factory=list)
    input_block_chans: List[int] = field(default_factory=list)
    # spatial downsample factor at which each input block's output lives (1,2,4,8)
    input_block_ds: List[int] = field(default_factory=list)


def unet_plan(cfg: UNetConfig, with_decoder: bool = True) -> UNetPlan:
    """Mirror of the constructor loops in `openaimodel.py:544-732` (UNetModel) and
    `cldm/cldm.py:138-279` (ControlNet encoder copy; `with_decoder=False`)."""
    mc = cfg.model_channels
    plan = UNetPlan(cfg)
    plan.input_blocks.append([Block("conv_in", "input_blocks.0.0", cfg.in_channels, mc)])
    chans = [mc]
    dss = [1]
    ch, ds = mc, 1
    idx = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [Block("res", f"input_blocks.{idx}.0", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(Block("attn", f"input_blocks.{idx}.1", ch, ch, cfg.num_heads))
            plan.input_blocks.append(layers)
            chans.append(ch)
            dss.append(ds)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            plan.input_blocks.append([Block("down", f"input_blocks.{idx}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
            dss.append(ds)
            idx += 1
    plan.input_block_chans = list(chans)
    plan.input_block_ds = list(dss)
    plan.middle_block = [
        Block("res", "middle_block.0", ch, ch),
        Block("attn", "middle_block.1", ch, ch, cfg.num_heads),
        Block("res", "middle_block.2", ch, ch),
    ]
    if not with_decoder:
        return plan
    stack = list(chans)
    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = stack.pop()
            layers = [Block("res", f"output_blocks.{oidx}.0", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(Block("attn", f"output_blocks.{oidx}.1", ch, ch, cfg.num_heads))
            if level and i == cfg.num_res_blocks:
                layers.append(Block("up", f"output_blocks.{oidx}.{len(layers)}", ch, ch))
                ds //= 2
            plan.output_blocks.append(layers)
            oidx += 1
    return plan


HINT_BLOCK_CHANNELS = [(None, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1),
                       (96, 256, 2), (256, None, 1)]  # (cin, cout, stride); `cldm/cldm.py:147-163`


def hint_block_convs(cfg: UNetConfig) -> List[Tuple[str, int, int, int]]:
    out = []
    for i, (ci, co, s) in enumerate(HINT_BLOCK_CHANNELS):
        ci = cfg.hint_channels if ci is None else ci
        co = cfg.model_channels if co is None else co
        out.append((f"input_hint_block.{2 * i}", ci, co, s))
    return out


# ----------------------------------------------------------------------------------------
# parameter inventory
# ----------------------------------------------------------------------------------------

Spec = "OrderedDict[str, Tuple[int, ...]]"


def _conv(spec, name, cin, cout, k):
    spec[f"{name}.weight"] = (cout, cin, k, k)
    spec[f"{name}.bias"] = (cout,)


def _lin(spec, name, cin, cout, bias=True):
    spec[f"{name}.weight"] = (cout, cin)
    if bias:
        spec[f"{name}.bias"] = (cout,)


def _norm(spec, name, c):
    spec[f"{name}.weight"] = (c,)
    spec[f"{name}.bias"] = (c,)


def _res(spec, b: Block, emb: int):
    """ResBlock parameters (`openaimodel.py:200-240`)."""
    _norm(spec, f"{b.name}.in_layers.0", b.cin)
    _conv(spec, f"{b.name}.in_layers.2", b.cin, b.cout, 3)
    _lin(spec, f"{b.name}.emb_layers.1", emb, b.cout)
    _norm(spec, f"{b.name}.out_layers.0", b.cout)
    _conv(spec, f"{b.name}.out_layers.3", b.cout, b.cout, 3)
    if b.cin != b.cout:
        _conv(spec, f"{b.name}.skip_connection", b.cin, b.cout, 1)


def _attn(spec, b: Block, ctx: int):
    """SpatialTransformer parameters (`ldm/modules/attention.py:397-429`, depth 1,
    BasicTransformerBlock `:360-375`, CrossAttention `:154-179`, GEGLU FeedForward `:49-76`)."""
    c = b.cin
    _norm(spec, f"{b.name}.norm", c)
    _conv(spec, f"{b.name}.proj_in", c, c, 1)
    t = f"{b.name}.transformer_blocks.0"
    for a, kd in (("attn1", c), ("attn2", ctx)):
        _lin(spec, f"{t}.{a}.to_q", c, c, bias=False)
        _lin(spec, f"{t}.{a}.to_k", kd, c, bias=False)
        _lin(spec, f"{t}.{a}.to_v", kd, c, bias=False)
        _lin(spec, f"{t}.{a}.to_out.0", c, c)
    _lin(spec, f"{t}.ff.net.0.proj", c, 8 * c)
    _lin(spec, f"{t}.ff.net.2", 4 * c, c)
    for n in ("norm1", "norm2", "norm3"):
        _norm(spec, f"{t}.{n}", c)
    _conv(spec, f"{b.name}.proj_out", c, c, 1)


def _blocks(spec, blocks: List[Block], cfg: UNetConfig):
    for b in blocks:
        if b.kind == "conv_in":
            _conv(spec, b.name, b.cin, b.cout, 3)
        elif b.kind == "res":
            _res(spec, b, cfg.time_embed_dim)
        elif b.kind == "attn":
            _attn(spec, b, cfg.context_dim)
        elif b.kind == "down":
            _conv(spec, f"{b.name}.op", b.cin, b.cout, 3)
        elif b.kind == "up":
            _conv(spec, f"{b.name}.conv", b.cin, b.cout, 3)
        else:
            raise ValueError(b.kind)


def param_spec_unet(cfg: UNetConfig = UNET_SD15):
    """ControlledUnetModel/UNetModel parameters, reference naming (`openaimodel.py:528-732`)."""
    spec = OrderedDict()
    plan = unet_plan(cfg)
    _lin(spec, "time_embed.0", cfg.model_channels, cfg.time_embed_dim)
    _lin(spec, "time_embed.2", cfg.time_embed_dim, cfg.time_embed_dim)
    for blocks in plan.input_blocks:
        _blocks(spec, blocks, cfg)
    _blocks(spec, plan.middle_block, cfg)
    for blocks in plan.output_blocks:
        _blocks(spec, blocks, cfg)
    _norm(spec, "out.0", cfg.model_channels)
    _conv(spec, "out.2", cfg.model_channels, cfg.out_channels, 3)
    return spec


def param_spec_controlnet(cfg: UNetConfig = UNET_SD15):
    """ControlNet parameters, reference naming (`cldm/cldm.py:131-282`)."""
    spec = OrderedDict()
    plan = unet_plan(cfg, with_decoder=False)
    _lin(spec, "time_embed.0", cfg.model_channels, cfg.time_embed_dim)
    _lin(spec, "time_embed.2", cfg.time_embed_dim, cfg.time_embed_dim)
    for blocks in plan.input_blocks:
        _blocks(spec, blocks, cfg)
    for i, ch in enumerate(plan.input_block_chans):
        _conv(spec, f"zero_convs.{i}.0", ch, ch, 1)
    for name, ci, co, _ in hint_block_convs(cfg):
        _conv(spec, name, ci, co, 3)
    _blocks(spec, plan.middle_block, cfg)
    ch = plan.input_block_chans[-1]
    _conv(spec, "middle_block_out.0", ch, ch, 1)
    return spec


def vae_plan(cfg: VAEConfig = VAE_SD15):
    """Decoder structure (`ldm/modules/diffusionmodules/model.py:546-616`): returns
    (block_in, [(level, [(cin,cout)...], has_upsample)]) in execution order (highest level first)."""
    nres = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[nres - 1]
    levels = []
    bi = block_in
    for i_level in reversed(range(nres)):
        bo = cfg.ch * cfg.ch_mult[i_level]
        blocks = []
        for _ in range(cfg.num_res_blocks + 1):
            blocks.append((bi, bo))
            bi = bo
        levels.append((i_level, blocks, i_level != 0))
    return block_in, levels


def _vae_res(spec, name, cin, cout):
    _norm(spec, f"{name}.norm1", cin)
    _conv(spec, f"{name}.conv1", cin, cout, 3)
    _norm(spec, f"{name}.norm2", cout)
    _conv(spec, f"{name}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, f"{name}.nin_shortcut", cin, cout, 1)


def param_spec_vae(cfg: VAEConfig = VAE_SD15):
    """`first_stage_model.*` parameters on the decode path: post_quant_conv + decoder.*
    (AutoencoderKL itself is absent from the reference tree; naming follows upstream)."""
    spec = OrderedDict()
    _conv(spec, "post_quant_conv", cfg.embed_dim, cfg.z_channels, 1)
    block_in, levels = vae_plan(cfg)
    d = "decoder"
    _conv(spec, f"{d}.conv_in", cfg.z_channels, block_in, 3)
    _vae_res(spec, f"{d}.mid.block_1", block_in, block_in)
    _norm(spec, f"{d}.mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        _conv(spec, f"{d}.mid.attn_1.{n}", block_in, block_in, 1)
    _vae_res(spec, f"{d}.mid.block_2", block_in, block_in)
    last = block_in
    for i_level, blocks, has_up in levels:
        for j, (ci, co) in enumerate(blocks):
            _vae_res(spec, f"{d}.up.{i_level}.block.{j}", ci, co)
            last = co
        if has_up:
            _conv(spec, f"{d}.up.{i_level}.upsample.conv", last, last, 3)
    _norm(spec, f"{d}.norm_out", last)
    _conv(spec, f"{d}.conv_out", last, cfg.out_ch, 3)
    return spec


# checkpoint namespaces of `control_sd15_canny.pth`
NS_UNET = "model.diffusion_model."
NS_CONTROL = "control_model."
NS_VAE = "first_stage_model."


def param_spec_full(ucfg: UNetConfig = UNET_SD15, vcfg: VAEConfig = VAE_SD15):
    spec = OrderedDict()
    for ns, s in ((NS_UNET, param_spec_unet(ucfg)), (NS_CONTROL, param_spec_controlnet(ucfg)),
                  (NS_VAE, param_spec_vae(vcfg))):
        for k, v in s.items():
            spec[ns + k] = v
    return spec


# ----------------------------------------------------------------------------------------
# synthetic weights
# ----------------------------------------------------------------------------------------


def _seed_for(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


@dataclass(frozen=True)
class ClipConfig:
    """CLIP text transformer behind `FrozenCLIPEmbedder` (`ldm/modules/encoders/modules.py:90-141`):
    openai/clip-vit-large-patch14 text tower (HuggingFace `CLIPTextConfig` of that checkpoint)."""
    vocab: int = 49408
    positions: int = 77
    width: int = 768
    layers: int = 12
    heads: int = 12
    ffn: int = 3072


CLIP_SD15 = ClipConfig()
CLIP_TINY = ClipConfig(vocab=1000, positions=77, width=64, layers=2, heads=4, ffn=128)
NS_CLIP = "cond_stage_model.transformer.text_model."


def param_spec_clip(cfg: ClipConfig = CLIP_SD15) -> "OrderedDict[str, tuple]":
    """Tensor names below `text_model.` and shapes of the HuggingFace CLIPTextModel state dict."""
    w, f = cfg.width, cfg.ffn
    spec = OrderedDict()
    spec["embeddings.token_embedding.weight"] = (cfg.vocab, w)
    spec["embeddings.position_embedding.weight"] = (cfg.positions, w)
    for i in range(cfg.layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec[p + f"self_attn.{n}.weight"] = (w, w)
            spec[p + f"self_attn.{n}.bias"] = (w,)
        spec[p + "layer_norm1.weight"] = (w,)
        spec[p + "layer_norm1.bias"] = (w,)
        spec[p + "mlp.fc1.weight"] = (f, w)
        spec[p + "mlp.fc1.bias"] = (f,)
        spec[p + "mlp.fc2.weight"] = (w, f)
        spec[p + "mlp.fc2.bias"] = (w,)
        spec[p + "layer_norm2.weight"] = (w,)
        spec[p + "layer_norm2.bias"] = (w,)
    spec["final_layer_norm.weight"] = (w,)
    spec["final_layer_norm.bias"] = (w,)
    return spec


def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Deterministic stand-in for one checkpoint tensor (fp32, CPU).

    Norm scales ~ 1 + 0.1 N(0,1), biases ~ 0.02 N(0,1), matrices ~ N(0, 1/fan_in) * gain.  The
    reference's `zero_module` tensors are drawn like every other tensor (with all-zero
    zero-convs every network output is exactly 0)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(seed, name))
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 1:
        is_norm = any(t in name for t in (".norm", "in_layers.0", "out_layers.0", "out.0", "norm_out", "layer_norm"))
        if leaf == "weight" and is_norm:
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.02 * torch.randn(shape, generator=g)
    if "_embedding.weight" in name:          # CLIP token / position tables: entries of order 0.02-ish like the real ones
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5


def synth_state_dict(spec, seed: int = 0, prefix: str = "") -> Dict[str, torch.Tensor]:
    return OrderedDict((k, synth_tensor(prefix + k, shp, seed)) for k, shp in spec.items())


def count_params(spec) -> int:
    n = 0
    for shp in spec.values():
        p = 1
        for d in shp:
            p *= d
        n += p
    return n
