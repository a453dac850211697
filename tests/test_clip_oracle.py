"""CPU: the CLIP text-transformer restatement (oracle/clip_oracle.py) against the golden vectors produced by the installed
HuggingFace `transformers.CLIPTextModel` (tests/golden/make_golden_clip.py), and host-side checks of the mirror class."""
import os

import numpy as np
import pytest
import torch

from oracle.clip_oracle import clip_text_forward
from stablediffusioneo_amd import spec as S

GOLD = os.path.join(os.path.dirname(__file__), "golden", "clip_tiny.npz")


def load_gold():
    z = np.load(GOLD)
    sd = {k[2:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w:")}
    return sd, torch.from_numpy(z["tokens"].astype(np.int64)), torch.from_numpy(z["last_hidden_state"])


def test_oracle_matches_transformers_golden():
    sd, tokens, want = load_gold()
    got = clip_text_forward(sd, tokens, S.CLIP_TINY.heads)
    assert got.shape == want.shape
    # fp32 on both sides; the only differences are summation orders
    assert float((got - want).abs().max()) < 2e-5


def test_golden_covers_the_declared_tensors():
    sd, _, _ = load_gold()
    spec = S.param_spec_clip(S.CLIP_TINY)
    assert set(sd) == set(spec)
    for k, shp in spec.items():
        assert tuple(sd[k].shape) == tuple(shp), k


def test_causality_and_padding_independence():
    """token t's output depends only on tokens <= t (causal mask): changing the padding after EOS changes nothing before it"""
    sd, tokens, _ = load_gold()
    a = clip_text_forward(sd, tokens, S.CLIP_TINY.heads)
    t2 = tokens.clone()
    t2[0, 20:] = 5
    b = clip_text_forward(sd, t2, S.CLIP_TINY.heads)
    assert torch.equal(a[0, :20], b[0, :20]) and not torch.equal(a[0, 20:], b[0, 20:])
    assert torch.equal(a[1], b[1])


def test_sd15_param_count():
    n = S.count_params(S.param_spec_clip(S.CLIP_SD15))
    assert n == 123_060_480          # openai/clip-vit-large-patch14 text tower (without the position_ids buffer)


def test_hash_tokenizer_is_deterministic():
    from stablediffusioneo_amd.ldm.modules.encoders.modules import HashTokenizer
    tk = HashTokenizer(vocab=49408, max_length=77)
    a = tk(["a photo of a cat", "a photo of a cat, best quality"])
    b = tk(["a photo of a cat", "a photo of a cat, best quality"])
    assert a.shape == (2, 77) and np.array_equal(a, b)
    assert a[0, 0] == 49406 and a[0, 6] == 49407 and (a[0, 6:] == 49407).all()      # BOS, 5 words, EOS + padding
    assert (a[0, 1:6] == a[1, 1:6]).all() and a[1, 6] != 49407
