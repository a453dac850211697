"""Shared synthetic inputs for the parity tests, the golden-fixture generator, smoke() and bench.py.

Everything is drawn from the torch CPU generator so the oracle (CPU), the HIP path (GPU box) and
`tests/golden/make_golden.py` (build container, reference modules) see identical tensors.
Seeds follow SURVEY.md 8(d): x_T seed 2946901 (the reference's seed, `compute_score_torch.py:37`),
cond/uncond context seeds 1/2, hint = seeded {0,1} edge map with 3 identical channels
(mimics `HWC3(Canny)`, `canny2image_torch.py:33-36`)."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
X_T_SEED = 2946901


def randn(shape, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(shape, generator=g)


def make_hint(n, H, W, seed=3, p=0.08):
    """(n,3,H,W) float32 in {0,1}: Bernoulli(p) edge map, three identical channels."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    m = (torch.rand((n, 1, H, W), generator=g) < p).float()
    return m.expand(n, 3, H, W).contiguous()


def make_inputs(n, h, w, ctx_dim=768, ctx_len=77, in_ch=4, x_seed=X_T_SEED, ctx_seed=1, hint_seed=3):
    x = randn((n, in_ch, h, w), x_seed)
    ctx = randn((n, ctx_len, ctx_dim), ctx_seed)
    hint = make_hint(n, 8 * h, 8 * w, hint_seed)
    return x, ctx, hint
