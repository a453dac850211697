"""GPU parity of the CLIP text transformer through the C ABI (`sdeo_clip_*`):
  * tiny configuration with the golden weights: against `last_hidden_state` of HuggingFace `transformers.CLIPTextModel`
    itself (tests/golden/clip_tiny.npz) and against the oracle;
  * the SD-1.5 configuration (openai/clip-vit-large-patch14 text tower: 12 layers, width 768) with seeded synthetic weights
    against the oracle on the same tokens;
  * the FrozenCLIPEmbedder mirror end to end.
Tolerance: fp16 storage / fp32 accumulation against an fp32 reference -- max |err| <= 2e-2 * max |ref| (measured ~2e-3)."""
import numpy as np
import pytest
import torch

from oracle.clip_oracle import clip_text_forward
from stablediffusioneo_amd import spec as S
from tests.test_clip_oracle import load_gold

pytestmark = pytest.mark.gpu
REL = 2e-2


def rel_err(got, want):
    return float((got.cpu().float() - want).abs().max() / want.abs().max())


def test_tiny_matches_transformers_golden():
    from stablediffusioneo_amd.runtime import ClipRuntime
    sd, tokens, want = load_gold()
    rt = ClipRuntime(S.CLIP_TINY).load_state_dict(sd, strict=True).configure(2)
    got = rt.encode(tokens)
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert rel_err(got, want) < REL
    assert rel_err(got, clip_text_forward(sd, tokens, S.CLIP_TINY.heads)) < REL
    assert torch.equal(got, rt.encode(tokens))                       # deterministic


def test_prefixed_names_and_batch_change():
    from stablediffusioneo_amd.runtime import ClipRuntime
    sd, tokens, want = load_gold()
    full = {S.NS_CLIP + k: v for k, v in sd.items()}                 # names as in an SD checkpoint
    full["cond_stage_model.transformer.text_model.embeddings.position_ids"] = torch.arange(77)[None]
    full["model.diffusion_model.out.2.bias"] = torch.zeros(4)       # foreign tensors are ignored when not strict
    rt = ClipRuntime(S.CLIP_TINY).load_state_dict(full)
    one = rt.encode(tokens[1:2])                                     # reconfigures to batch 1
    assert rel_err(one, want[1:2]) < REL
    three = rt.encode(torch.cat([tokens, tokens[:1]]))
    assert torch.equal(three[0], three[2]) and rel_err(three[:2], want) < REL


def test_sd15_config_matches_oracle():
    from stablediffusioneo_amd.runtime import ClipRuntime
    cfg = S.CLIP_SD15
    spec = S.param_spec_clip(cfg)
    sd = {k: S.synth_tensor(S.NS_CLIP + k, shp, 3).half().float() for k, shp in spec.items()}
    g = torch.Generator().manual_seed(11)
    tokens = torch.randint(0, cfg.vocab - 2, (2, cfg.positions), generator=g)
    tokens[:, 0] = cfg.vocab - 2
    tokens[0, 12:] = cfg.vocab - 1
    want = clip_text_forward(sd, tokens, cfg.heads)
    rt = ClipRuntime(cfg).load_state_dict(sd, strict=True).configure(2)
    got = rt.encode(tokens)
    assert torch.isfinite(got).all() and rel_err(got, want) < REL
    assert 200e6 < rt.device_bytes() < 400e6                        # ~246 MB of fp16 weights + activations


def test_frozen_clip_embedder_mirror():
    from stablediffusioneo_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    from stablediffusioneo_amd.runtime import ClipRuntime
    sd, _, _ = load_gold()
    enc = FrozenCLIPEmbedder(config=S.CLIP_TINY, runtime=ClipRuntime(S.CLIP_TINY).load_state_dict(sd), allow_hash_tokenizer=True)
    prompts = ["a bird, best quality, extremely detailed", "lowres, bad anatomy"]
    z = enc.encode(prompts)
    assert z.shape == (2, 77, 64) and z.dtype == torch.float32 and z.is_cuda
    want = clip_text_forward(sd, enc.tokenize(prompts), S.CLIP_TINY.heads)
    assert rel_err(z, want) < REL
    with pytest.raises(NotImplementedError):
        FrozenCLIPEmbedder(layer="pooled", config=S.CLIP_TINY, runtime=enc.transformer, allow_hash_tokenizer=True)
