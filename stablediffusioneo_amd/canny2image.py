"""The `hackathon` pipeline of `canny2image_torch.py:18-71` / `canny2image_TRT.py:18-92` on the libsdeo path:
same `initialize()` / `process(...)` signature and semantics (Canny hint -> text conditioning -> DDIM loop with
classifier-free guidance -> VAE decode -> uint8 HWC images).

The two stages around the loop are injectable:
  * the Canny detector: `apply_canny(img, low, high) -> HxW uint8`; default = the HIP `CannyDetector`
    (`stablediffusioneo_amd/annotator/canny`, csrc/canny.hip), whose `control_hint` keeps the hint on the GPU;
  * the CLIP text encoder (`FrozenCLIPEmbedder`): `text_encoder(prompts) -> (B,77,768)`; the default is a
    deterministic synthetic embedding (seeded by the prompt text) so the pipeline is runnable without weights.
"""
from __future__ import annotations

import hashlib
import random

import numpy as np
import torch

from . import spec as S
from .annotator.util import HWC3, resize_image, target_size  # noqa: F401  (`from annotator.util import resize_image, HWC3`, canny2image_torch.py:6)
from .cldm.ddim_hacked import DDIMSampler
from .cldm.model import create_model

save_memory = False     # `config.py:1`


def synthetic_text_encoder(prompts, length=77, dim=768):
    out = []
    for p in prompts:
        g = torch.Generator(device="cpu")
        g.manual_seed(int.from_bytes(hashlib.sha256(p.encode()).digest()[:7], "little"))
        out.append(torch.randn((length, dim), generator=g))
    return torch.stack(out)


def _default_canny():
    from .annotator.canny import CannyDetector     # `annotator/canny/__init__.py:4-6` on the HIP path (csrc/canny.hip)
    return CannyDetector()


class hackathon():

    def initialize(self, weights="synthetic:0", config="sd15", apply_canny=None, text_encoder=None):
        """text_encoder: None = `synthetic_text_encoder` (seeded stand-in contexts); "clip:<tokenizer dir>" = the
        FrozenCLIPEmbedder mirror on the HIP path (weights from the same source as the UNet's: synthetic seed or the
        checkpoint's `cond_stage_model.transformer.text_model.*`); bare "clip" is accepted only with synthetic weights (the
        tokenizer is then the crc32 stand-in, which is meaningless next to real weights); or any callable(prompts) ->
        (B, 77, 768) tensor."""
        self.apply_canny = apply_canny or _default_canny()
        if isinstance(text_encoder, str) and text_encoder.split(":")[0] == "clip":
            from . import spec as S
            from .ldm.modules.encoders.modules import FrozenCLIPEmbedder
            tok_dir = text_encoder.split(":", 1)[1] if ":" in text_encoder else None
            synthetic = isinstance(weights, str) and weights.startswith("synthetic")
            text_encoder = FrozenCLIPEmbedder(version=tok_dir, config=S.CLIP_TINY if config == "tiny" else S.CLIP_SD15,
                                              allow_hash_tokenizer=synthetic and tok_dir is None)
        self.text_encoder = text_encoder or synthetic_text_encoder
        self.model = create_model(config, cond_stage_model=self.text_encoder)
        if isinstance(weights, str) and weights.startswith("synthetic"):
            seed = int(weights.split(":")[1]) if ":" in weights else 0
            self.model.rt.load_synthetic(seed)
            if hasattr(self.text_encoder, "transformer"):
                self.text_encoder.transformer.load_synthetic(seed)
        elif isinstance(weights, dict):
            self.model.load_state_dict(weights)
        else:
            from .cldm.model import load_state_dict
            self.model.load_state_dict(load_state_dict(weights, location="cuda"))
        self.ddim_sampler = DDIMSampler(self.model)
        return self

    def process(self, input_image, prompt, a_prompt, n_prompt, num_samples, image_resolution, ddim_steps, guess_mode,
                strength, scale, seed, eta, low_threshold, high_threshold, x_T=None):
        with torch.no_grad():
            img = resize_image(HWC3(input_image), image_resolution)
            H, W, C = img.shape
            device = self.model.device
            if hasattr(self.apply_canny, "control_hint"):
                # edges -> HWC3 -> /255 -> CHW on the GPU (`canny2image_torch.py:33-38` without the host round trip)
                control = self.apply_canny.control_hint(img, low_threshold, high_threshold).to(device)
                control = torch.stack([control for _ in range(num_samples)], dim=0).contiguous()
            else:
                detected_map = HWC3(self.apply_canny(img, low_threshold, high_threshold))
                control = torch.from_numpy(detected_map.copy()).float().to(device) / 255.0
                control = torch.stack([control for _ in range(num_samples)], dim=0)
                control = control.permute(0, 3, 1, 2).contiguous()
            if seed == -1:
                seed = random.randint(0, 65535)
            random.seed(seed)
            np.random.seed(seed)
            torch.manual_seed(seed)       # pytorch_lightning.seed_everything (`canny2image_torch.py:42`)
            cond = {"c_concat": [control],
                    "c_crossattn": [self.model.get_learned_conditioning([prompt + ", " + a_prompt] * num_samples)]}
            un_cond = {"c_concat": None if guess_mode else [control],
                       "c_crossattn": [self.model.get_learned_conditioning([n_prompt] * num_samples)]}
            shape = (4, H // 8, W // 8)
            # `canny2image_torch.py:54`: guess-mode scales 0.825**(12-i)
            self.model.control_scales = ([strength * (0.825 ** float(12 - i)) for i in range(13)] if guess_mode
                                         else ([strength] * 13))
            samples, intermediates = self.ddim_sampler.sample(ddim_steps, num_samples, shape, cond, verbose=False, eta=eta,
                                                              unconditional_guidance_scale=scale,
                                                              unconditional_conditioning=un_cond, x_T=x_T)
            x_samples = self.model.decode_first_stage_uint8(samples).cpu().numpy()
            results = [x_samples[i] for i in range(num_samples)]
        return results
