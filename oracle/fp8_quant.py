"""CPU restatement of libsdeo's fp8 weight pack (csrc/elementwise.hip: quantize_fp8_rows_kernel).  TEST INFRASTRUCTURE ONLY.

BASELINE configs[4] stores the UNet / ControlNet matrices as OCP e4m3fn with one scale per output channel; the reference has no
fp8 code of its own (its precision switch is a TensorRT builder flag, `onnx2trt_static_plugin.py:40-42`), so the format is
this build's and is pinned here against torch's own float8_e4m3fn cast (round to nearest even):

    scale[r] = the smallest power of two >= 2^-15 with max|w[r]| / scale[r] <= 448     (1 for an all-zero row)
    code     = e4m3fn(w / scale)          dequantised weight = float(code) * scale   (exact in fp16)

`quantised_state_dict` applies it to a reference state dict the way the library does: every UNet / ControlNet matrix per output
row after rounding to fp16; the three Linear layers of a BasicTransformerBlock that consume a LayerNorm are quantised AFTER the
LayerNorm scale is folded into them (that is the matrix the network streams), which in reference terms is W_eff = Q(W * gamma) /
gamma with the LayerNorm left as it is."""
from __future__ import annotations

import re

import torch


def quantize_rows(w: torch.Tensor):
    """w: [rows][cols] float (values already on the fp16 grid) -> (codes uint8 [rows][cols], scale f32 [rows], dequantised f32)."""
    w = w.float()
    amax = w.abs().amax(dim=1)
    m, x = torch.frexp(amax)
    e = torch.where(m <= 0.875, x - 9, x - 8).clamp(min=-15)       # code * scale must stay on the fp16 grid
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), e), torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    codes = q.view(torch.uint8).clone()
    over = (q.float() * scale[:, None]).abs() > 65504.0      # fp16's top binade may round past fp16's maximum: next code down
    codes[over] -= 1
    return codes, scale, codes.view(torch.float8_e4m3fn).float() * scale[:, None]


def quantize_mx(x: torch.Tensor):
    """Block-scaled fp8 pack (csrc/elementwise.hip quantize_mx_kernel): x [rows][cols] (values on the fp16 grid), cols % 32 == 0 ->
    (codes uint8 [rows][cols], e8m0 scale bytes uint8 [rows][cols / 32], dequantised f32).  Per block of 32 consecutive elements of a
    row: scale = 2^e, the smallest power of two with max|x| / scale <= 448 (e = 0 for an all-zero block), stored as e + 127; codes =
    e4m3fn(x / scale), round to nearest even (torch's cast).  The MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) multiplies the decoded
    codes and the two block scales exactly and accumulates in fp32."""
    rows, cols = x.shape
    assert cols % 32 == 0
    b = x.float().reshape(rows, cols // 32, 32)
    amax = b.abs().amax(dim=2)
    m, ex = torch.frexp(amax)
    e = torch.where(amax > 0, torch.where(m <= 0.875, ex - 9, ex - 8), torch.zeros_like(ex))
    scale = torch.ldexp(torch.ones_like(amax), e)
    q = (b / scale[:, :, None]).to(torch.float8_e4m3fn)
    codes = q.view(torch.uint8).reshape(rows, cols).clone()
    deq = (q.float() * scale[:, :, None]).reshape(rows, cols)
    return codes, (e + 127).to(torch.uint8), deq


def _q2d(w: torch.Tensor) -> torch.Tensor:
    """quantise a Linear [out][in] or conv [out][in][kh][kw] weight per output channel (library layout: [out][kh][kw][in])"""
    w16 = w.to(torch.float16).float()
    if w.dim() == 4:
        o = w16.shape[0]
        deq = quantize_rows(w16.permute(0, 2, 3, 1).reshape(o, -1))[2]
        return deq.reshape(o, w.shape[2], w.shape[3], w.shape[1]).permute(0, 3, 1, 2).contiguous()
    return quantize_rows(w16)[2]


def quantised_state_dict(sd):
    """The fp32 state dict (reference names) the oracle has to run to reproduce the fp8-weight network: UNet / ControlNet matrices
    replaced by their dequantised values; biases, norms and the VAE untouched."""
    out = dict(sd)
    for k, w in sd.items():
        if not k.startswith(("model.diffusion_model.", "control_model.")) or w.dim() < 2:
            continue
        m = re.match(r"(.*\.transformer_blocks\.0)\.(attn1\.to_[qkv]|attn2\.to_q|ff\.net\.0\.proj)\.weight$", k)
        if m:
            norm = {"attn1": "norm1", "attn2": "norm2", "ff": "norm3"}[m.group(2).split(".")[0]]
            gamma = sd[f"{m.group(1)}.{norm}.weight"].float()
            wf = (w.to(torch.float16).float() * gamma[None, :]).to(torch.float16).float()       # the fold: fp16(W16 * gamma)
            deq = quantize_rows(wf)[2]
            out[k] = torch.where(gamma[None, :] != 0, deq / gamma[None, :], torch.zeros_like(deq))
        else:
            out[k] = _q2d(w)
    return out
