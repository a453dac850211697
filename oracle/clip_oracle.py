"""CPU restatement (plain fp32 torch) of the CLIP text transformer behind `FrozenCLIPEmbedder`
(`ldm/modules/encoders/modules.py:123-141`).  TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline
and __graft_entry__.smoke() as the checker, never by the product path.

The algorithm lives in a third-party dependency of the reference that is not vendored in /root/reference:
HuggingFace `transformers` (`CLIPTextModel`; the reference's environment pins transformers==4.19.2, the container
has 5.15.0).  This file restates the published forward pass of `CLIPTextTransformer` with `hidden_act="quick_gelu"`
(`CLIPTextEmbeddings`, `CLIPEncoderLayer`, `CLIPAttention` with the causal mask, `CLIPMLP`, `final_layer_norm`) and is
pinned by tests/golden/clip_tiny.npz, produced by running the installed `transformers.CLIPTextModel` itself
(tests/golden/make_golden_clip.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_text_forward(sd, tokens, heads: int, eps: float = 1e-5):
    """sd: tensors named as below `text_model.`; tokens: int64 [B, T] -> last_hidden_state fp32 [B, T, W]."""
    g = lambda k: sd[k].float()
    tok = g("embeddings.token_embedding.weight")
    pos = g("embeddings.position_embedding.weight")
    B, T = tokens.shape
    W = tok.shape[1]
    d = W // heads
    x = tok[tokens.long()] + pos[:T][None]
    mask = torch.full((T, T), float("-inf")).triu(1)          # key j > query t is masked
    layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(layers):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (W,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        q = F.linear(h, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias")) * d ** -0.5
        k = F.linear(h, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias"))
        v = F.linear(h, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias"))
        sp = lambda t: t.view(B, T, heads, d).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) + mask, dim=-1) @ sp(v)
        a = a.transpose(1, 2).reshape(B, T, W)
        x = x + F.linear(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (W,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        h = quick_gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
        x = x + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return F.layer_norm(x, (W,), g("final_layer_norm.weight"), g("final_layer_norm.bias"), eps)
