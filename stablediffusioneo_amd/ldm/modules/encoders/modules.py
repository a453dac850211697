"""`ldm/modules/encoders/modules.py` mirror: FrozenCLIPEmbedder on the HIP path.

The reference class (`ldm/modules/encoders/modules.py:90-141`) owns a HuggingFace tokenizer and `CLIPTextModel`, both
fetched by name from the hub.  Here the transformer runs through libsdeo.so (`sdeo_clip_*`); the tokenizer stays on the
host and is loaded from LOCAL files only (`version` = a directory holding vocab.json / merges.txt) -- there is no network
access at run time.  Without such files construction FAILS unless the caller opts in to `HashTokenizer`
(`allow_hash_tokenizer=True`: deterministic, NOT CLIP's BPE -- plumbing for synthetic weights only; loading real weights
next to it warns, because crc32 ids index the real embedding table at meaningless rows)."""
from __future__ import annotations

import os
import re
import warnings
import zlib
from typing import List, Optional, Sequence

import numpy as np
import torch

from .... import spec as S
from ....runtime import ClipRuntime


class AbstractEncoder:
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class HashTokenizer:
    """BOS + one id per lower-cased word / punctuation mark (crc32 mod vocab) + EOS, padded with EOS to max_length --
    the shape and padding convention of `CLIPTokenizer(..., padding="max_length", truncation=True)`
    (`ldm/modules/encoders/modules.py:124-125`), not its byte-pair vocabulary."""

    def __init__(self, vocab: int = 49408, max_length: int = 77):
        self.vocab, self.max_length = vocab, max_length
        self.bos, self.eos = vocab - 2, vocab - 1

    def __call__(self, texts: Sequence[str]) -> np.ndarray:
        out = np.full((len(texts), self.max_length), self.eos, dtype=np.int32)
        for i, t in enumerate(texts):
            words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", t.lower())
            ids = [self.bos] + [zlib.crc32(w.encode()) % (self.vocab - 2) for w in words][: self.max_length - 2] + [self.eos]
            out[i, : len(ids)] = ids
        return out


class FrozenCLIPEmbedder(AbstractEncoder):
    """Uses the CLIP transformer encoder for text; `forward(text)` returns `last_hidden_state` [B, 77, 768] (fp32, device)."""
    LAYERS = ["last"]      # the reference also lists "pooled" / "hidden"; cldm_v15 uses the default "last"

    def __init__(self, version: Optional[str] = None, device="cuda", max_length=77, freeze=True, layer="last", layer_idx=None,
                 config: S.ClipConfig = S.CLIP_SD15, runtime: Optional[ClipRuntime] = None, allow_hash_tokenizer: bool = False):
        if layer not in self.LAYERS:
            raise NotImplementedError(f"layer={layer!r}: only 'last' is built (what cldm_v15.yaml uses)")
        self.device = device
        self.max_length = max_length
        self.layer, self.layer_idx = layer, layer_idx
        self.config = config
        self.tokenizer = None
        if version is not None and os.path.isdir(version):
            from transformers import CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(version, local_files_only=True)
        if self.tokenizer is None:
            if not allow_hash_tokenizer:
                raise RuntimeError(
                    f"FrozenCLIPEmbedder: no local CLIP tokenizer at version={version!r} (a directory with vocab.json + merges.txt is "
                    f"needed; the reference's default 'openai/clip-vit-large-patch14' is a hub name and there is no network). "
                    f"Pass allow_hash_tokenizer=True to use the crc32 stand-in -- token ids are then NOT CLIP's BPE ids, which is only "
                    f"meaningful with synthetic weights.")
            self.tokenizer = HashTokenizer(config.vocab, max_length)
        self.transformer = runtime if runtime is not None else ClipRuntime(config)

    def freeze(self):
        return self

    def load_state_dict(self, sd, strict=False):
        if isinstance(self.tokenizer, HashTokenizer) and len(sd):
            warnings.warn("FrozenCLIPEmbedder: checkpoint weights loaded next to the HashTokenizer stand-in: prompts are hashed, not "
                          "BPE-tokenised, so the conditioning is meaningless for real weights (supply version=<tokenizer dir>)",
                          RuntimeWarning, stacklevel=2)
        self.transformer.load_state_dict(sd, strict=strict)
        return self

    def tokenize(self, text: List[str]) -> torch.Tensor:
        if isinstance(self.tokenizer, HashTokenizer):
            ids = self.tokenizer(text)
        else:
            ids = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                                 return_overflowing_tokens=False, padding="max_length", return_tensors="np")["input_ids"]
        return torch.from_numpy(np.asarray(ids, dtype=np.int64))

    def forward(self, text):
        if isinstance(text, str):
            text = [text]
        return self.transformer.encode(self.tokenize(list(text)))

    __call__ = forward

    def encode(self, text):
        return self(text)
