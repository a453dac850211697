"""Golden vectors for the CLIP text transformer: run the installed HuggingFace `transformers.CLIPTextModel` (the
third-party dependency `FrozenCLIPEmbedder` wraps, `ldm/modules/encoders/modules.py:100-101,131`) on a tiny seeded
configuration and store weights, token ids and `last_hidden_state`.

    python tests/golden/make_golden_clip.py      # writes tests/golden/clip_tiny.npz (fp16 weights to keep it small)
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stablediffusioneo_amd import spec as S      # noqa: E402


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    c = S.CLIP_TINY
    cfg = CLIPTextConfig(vocab_size=c.vocab, hidden_size=c.width, intermediate_size=c.ffn, num_hidden_layers=c.layers,
                         num_attention_heads=c.heads, max_position_embeddings=c.positions, hidden_act="quick_gelu",
                         bos_token_id=c.vocab - 2, eos_token_id=c.vocab - 1, pad_token_id=c.vocab - 1)
    torch.manual_seed(20250523)
    m = CLIPTextModel(cfg).eval()
    sd = {}
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if not torch.is_floating_point(v):
                continue
            k = k.split("text_model.")[-1]
            # spread the values a little (HF initialises LayerNorm to exactly 1 / 0 and biases to 0, which would hide
            # swapped or missing scale / bias terms), then round to fp16 so the fixture is what both sides load
            g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
            if v.dim() == 1:
                v = v + 0.05 * torch.randn(v.shape, generator=g)
            sd[k] = v.half().float()
        prefixed = any(n.startswith("text_model.") for n in m.state_dict())
        m.load_state_dict({("text_model." + k if prefixed else k): v for k, v in sd.items()}, strict=False)
        for k, v in m.state_dict().items():          # strict=False must not have skipped anything we perturbed
            if torch.is_floating_point(v):
                assert torch.equal(v, sd[k.split("text_model.")[-1]]), k
        g = torch.Generator().manual_seed(7)
        tokens = torch.randint(0, c.vocab - 2, (2, c.positions), generator=g)
        tokens[:, 0] = c.vocab - 2
        tokens[0, 9:] = c.vocab - 1          # a short prompt: EOS then padding, like the tokenizer's max_length padding
        tokens[1, 40:] = c.vocab - 1
        out = m(input_ids=tokens).last_hidden_state
    blob = {"w:" + k: v.half().numpy() for k, v in sd.items()}
    blob["tokens"] = tokens.numpy().astype(np.int32)
    blob["last_hidden_state"] = out.numpy().astype(np.float32)
    blob["transformers_version"] = np.array(transformers.__version__)
    path = os.path.join(ROOT, "tests", "golden", "clip_tiny.npz")
    np.savez_compressed(path, **blob)
    print(f"wrote {path}: {len(sd)} tensors, out {tuple(out.shape)}, |out| max {float(out.abs().max()):.3f}")


if __name__ == "__main__":
    main()
