"""Host-side mirror of the reference's model surface (`cldm/cldm.py`): ControlNet, ControlledUnetModel and
ControlLDM with the same names, call signatures and semantics, but every forward is ONE call into libsdeo
(the seam where the TensorRT variant calls `Engine.infer`, `cldm_trt/cldm.py:321-341,368-384`).

Also restates the attributes of the absent `LatentDiffusion` base class that the sampler and the pipeline
read: `num_timesteps`, `betas`, `alphas_cumprod`, `alphas_cumprod_prev`, `device`,
`parameterization`, `get_learned_conditioning`, `decode_first_stage`, `first_stage_model`, `model.diffusion_model`.
"""
from __future__ import annotations

import types
from typing import Callable, List, Optional

import numpy as np
import torch

from .. import spec as S
from ..runtime import CONTEXT_CACHED, HINT_CACHED, SdeoRuntime


def make_beta_schedule_linear(n_timestep, linear_start, linear_end):
    """`ldm/modules/diffusionmodules/util.py:21-25` ("linear": linspace in sqrt space, float64)."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


class ControlNet:
    """`cldm/cldm.py:48-305`.  forward(x, hint, timesteps, context) -> list of 13 NCHW fp32 tensors."""

    def __init__(self, runtime: SdeoRuntime):
        self.rt = runtime

    def forward(self, x, hint, timesteps, context, **kwargs):
        self.rt.configure(x.shape[0], x.shape[2], x.shape[3])
        return self.rt.controlnet(x, hint, timesteps, context)

    __call__ = forward


class ControlledUnetModel:
    """`cldm/cldm.py:22-45`.  Like the reference, `control` (a list) is CONSUMED by the call (`:35,41` pop())."""

    def __init__(self, runtime: SdeoRuntime):
        self.rt = runtime

    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        self.rt.configure(x.shape[0], x.shape[2], x.shape[3])
        ctrl = None
        if control is not None:
            ctrl = list(control)
            del control[:]
        return self.rt.unet(x, timesteps, context, control=ctrl, only_mid_control=only_mid_control)

    __call__ = forward


class AutoencoderKLDecoder:
    """decode side of `ldm.models.autoencoder.AutoencoderKL` (absent from the reference tree)."""

    def __init__(self, runtime: SdeoRuntime):
        self.rt = runtime

    def decode(self, z):
        """z is ALREADY divided by scale_factor (upstream AutoencoderKL.decode contract)."""
        return self.rt.vae_decode(z * self.rt.vcfg.scale_factor)


class ControlLDM:
    """`cldm/cldm.py:308-435` on top of a libsdeo runtime."""

    def __init__(self, runtime: SdeoRuntime, schedule: S.ScheduleConfig = S.SCHEDULE_SD15,
                 cond_stage_model: Optional[Callable[[List[str]], torch.Tensor]] = None, only_mid_control: bool = False):
        self.rt = runtime
        self.device = runtime.device
        self.control_model = ControlNet(runtime)
        self.model = types.SimpleNamespace(diffusion_model=ControlledUnetModel(runtime))
        self.first_stage_model = AutoencoderKLDecoder(runtime)
        self.cond_stage_model = cond_stage_model
        self.only_mid_control = only_mid_control
        self.control_scales = [1.0] * 13
        self.control_key = "hint"
        self.channels = runtime.ucfg.in_channels
        self.scale_factor = runtime.vcfg.scale_factor
        # LatentDiffusion.register_schedule (upstream semantics; file absent from the reference)
        self.parameterization = schedule.parameterization
        self.num_timesteps = schedule.timesteps
        betas = make_beta_schedule_linear(schedule.timesteps, schedule.linear_start, schedule.linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        to_t = lambda a: torch.tensor(a, dtype=torch.float32, device=self.device)
        self.betas = to_t(betas)
        self.alphas_cumprod = to_t(ac)
        self.alphas_cumprod_prev = to_t(np.append(1.0, ac[:-1]))
        self.sqrt_alphas_cumprod = to_t(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = to_t(np.sqrt(1.0 - ac))

    # -- nn.Module-ish conveniences the pipeline scripts call
    def cuda(self):
        return self

    def cpu(self):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=False):
        self.rt.load_state_dict(sd, strict=strict)
        if hasattr(self.cond_stage_model, "load_state_dict"):      # FrozenCLIPEmbedder mirror: cond_stage_model.transformer.*
            self.cond_stage_model.load_state_dict({k: v for k, v in sd.items() if "text_model." in k}, strict=False)
        return self

    def low_vram_shift(self, is_diffusing):
        """`cldm/cldm.py:425-435` moves sub-models between host and device to fit small GPUs; with 288 GB of
        HBM everything stays resident."""
        return None

    # -- conditioning
    def get_learned_conditioning(self, prompts):
        if self.cond_stage_model is None:
            raise RuntimeError("no cond_stage_model: pass the FrozenCLIPEmbedder mirror "
                               "(stablediffusioneo_amd.ldm.modules.encoders.modules) or any callable(prompts)->(B,77,768) tensor")
        return self.cond_stage_model(prompts).to(self.device)

    def get_unconditional_conditioning(self, N):
        return self.get_learned_conditioning([""] * N)

    # -- the hot call
    def q_sample(self, x_start, t, noise=None):
        """Forward diffusion x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps, called by the sampler's mask / x0 branch
        (`cldm/ddim_hacked.py:154-157`).  `LatentDiffusion.q_sample` itself is absent from the reference tree: upstream
        SD-v1 semantics, parity unpinned."""
        if noise is None:
            noise = torch.randn_like(x_start)
        ext = lambda a: a.to(x_start.device)[t].reshape(-1, *([1] * (x_start.dim() - 1)))
        return ext(self.sqrt_alphas_cumprod) * x_start + ext(self.sqrt_one_minus_alphas_cumprod) * noise

    def apply_model(self, x_noisy, t, cond, *args, flags: int = 0, out=None, **kwargs):
        """`cldm/cldm.py:328-341`."""
        assert isinstance(cond, dict)
        cond_txt = torch.cat(cond["c_crossattn"], 1) if cond["c_crossattn"] is not None else None
        self.rt.configure(x_noisy.shape[0], x_noisy.shape[2], x_noisy.shape[3])
        if cond["c_concat"] is None:
            return self.rt.apply_model(x_noisy, None, t, cond_txt, None, self.only_mid_control, flags & CONTEXT_CACHED, out)
        hint = torch.cat(cond["c_concat"], 1)
        return self.rt.apply_model(x_noisy, hint, t, cond_txt, self.control_scales, self.only_mid_control, flags, out)

    def decode_first_stage(self, z):
        """`canny2image_torch.py:63-67`: z = 1/scale_factor * z; first_stage_model.decode(z)."""
        return self.rt.vae_decode(z)

    def decode_first_stage_uint8(self, z):
        """decode + the post-process of `canny2image_torch.py:68`, fused on the device: (B,H,W,3) uint8."""
        return self.rt.vae_decode(z, want_u8=True)[1]
