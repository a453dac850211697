"""Op-level Python wrappers over the C ABI (include/sdeo.h).  torch is used only to own device
memory and the current HIP stream; every computation happens inside libsdeo.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, cur_stream, ptr

_f = C.c_float
_i = C.c_int


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SdeoError("HIP ops need device tensors (no CPU fallback)")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def groupnorm_nhwc(x, gamma, beta, groups=32, eps=1e-5, swish=False):
    """x: (N,H,W,C) fp16 contiguous; gamma/beta fp32 (C,)."""
    lib = _lib.load()
    _need_cuda(x, gamma, beta)
    n, h, w, c = x.shape
    assert x.dtype == torch.float16 and x.is_contiguous()
    y = torch.empty_like(x)
    ws = _ws(lib.sdeo_groupnorm_workspace_bytes(_i(n), _i(h * w), _i(groups)), x.device)
    check(lib.sdeo_groupnorm_nhwc_f16(ptr(y), ptr(x), ptr(gamma), ptr(beta), _i(n), _i(h), _i(w), _i(c), _i(groups),
                                      _f(eps), _i(int(swish)), ptr(ws), cur_stream()), "groupnorm")
    return y


def krsc_from_oihw(w_oihw_f32, cin_pad=None):
    """fp32 OIHW (device) -> fp16 [O][R][S][Ipad]."""
    lib = _lib.load()
    _need_cuda(w_oihw_f32)
    o, i, r, s = w_oihw_f32.shape
    ip = cin_pad or ((i + 7) // 8) * 8
    y = torch.empty((o, r, s, ip), dtype=torch.float16, device=w_oihw_f32.device)
    check(lib.sdeo_oihw_f32_to_krsc_f16(ptr(y), ptr(w_oihw_f32.contiguous()), _i(o), _i(i), _i(r), _i(s), _i(ip),
                                        cur_stream()), "krsc")
    return y


def quantize_fp8_rows(w):
    """The library's fp8 weight pack on a [rows][cols] fp16 matrix: (codes uint8, scales fp32 [rows], dequantised fp16)."""
    lib = _lib.load()
    _need_cuda(w)
    rows = w.shape[0]
    wd = w.reshape(rows, -1).clone().contiguous()
    q = torch.empty(wd.shape, dtype=torch.uint8, device=w.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=w.device)
    check(lib.sdeo_debug_quantize_fp8_rows(ptr(wd), ptr(q), ptr(sc), _i(rows), _i(wd.shape[1]), cur_stream()), "quantize_fp8_rows")
    return q.reshape(w.shape), sc, wd.reshape(w.shape)


def quantize_mx(x):
    """block-scaled fp8 pack of an fp16 [rows][cols] matrix (cols % 32 == 0): (codes uint8 [rows][cols], e8m0 scales uint8 [rows][cols/32])"""
    lib = _lib.load()
    _need_cuda(x)
    rows, cols = x.shape
    assert x.dtype == torch.float16 and x.is_contiguous() and cols % 32 == 0
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows, cols // 32), dtype=torch.uint8, device=x.device)
    check(lib.sdeo_debug_quantize_mx(ptr(q), ptr(sc), ptr(x), _i(rows), _i(cols), cur_stream()), "quantize_mx")
    return q, sc


def gemm_mx(xq, xs, wq, ws, bias=None, res=None, act=0):
    """y[m][n] = sum_k dequant(x)[m][k] dequant(w)[n][k] (+bias)(+res) on block-scaled fp8 operands (quantize_mx), K % 128 == 0; fp16 out"""
    lib = _lib.load()
    _need_cuda(xq, wq)
    m, k = xq.shape
    n = wq.shape[0]
    assert k % 128 == 0 and wq.shape[1] == k and xs.shape == (m, k // 32) and ws.shape == (n, k // 32)
    y = torch.empty((m, n // 2 if act == 3 else n), dtype=torch.float16, device=xq.device)
    wsb = _ws(64 << 20, xq.device)
    check(lib.sdeo_debug_gemm_mx_f16(ptr(y), _i(y.shape[1]), ptr(xq), ptr(xs), ptr(wq), ptr(ws), ptr(bias), ptr(res),
                                     _i(res.stride(0) if res is not None else 0), _i(m), _i(n), _i(k), _i(act), ptr(wsb),
                                     C.c_size_t(wsb.numel()), cur_stream()), "gemm_mx")
    return y


def _arm_fp8(lib, w8):
    if w8 is not None:
        q, sc = w8
        assert q.dtype == torch.uint8 and q.is_contiguous() and sc.dtype == torch.float32
        lib.sdeo_debug_next_weights_fp8(ptr(q), ptr(sc))


def conv2d_nhwc(x, w_krsc, bias=None, bias2=None, res=None, stride=1, upsample2x=False, act=0, scale=1.0, w8=None):
    """x (N,H,W,Cin) fp16; w_krsc (Cout,k,k,Cin) fp16; returns (N,Ho,Wo,Cout) fp16.  w8 = (codes, scales): stream the fp8 copy."""
    lib = _lib.load()
    _need_cuda(x, w_krsc)
    n, h, w, cin = x.shape
    cout, k, _, cin_w = w_krsc.shape
    assert cin == cin_w and x.is_contiguous() and w_krsc.is_contiguous()
    hv, wv = (2 * h, 2 * w) if upsample2x else (h, w)
    pad = k // 2
    ho = (hv + 2 * pad - k) // stride + 1
    wo = (wv + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    args = (_i(n), _i(h), _i(w), _i(cin), _i(cout), _i(k), _i(stride), _i(int(upsample2x)))
    nb = lib.sdeo_conv2d_workspace_bytes(*args)
    ws = _ws(nb if w8 is None else max(nb, 64 << 20), x.device)
    _arm_fp8(lib, w8)
    check(lib.sdeo_conv2d_nhwc_f16(ptr(y), ptr(x), ptr(w_krsc), ptr(bias), ptr(bias2), ptr(res), *args, _i(act), _f(scale),
                                   ptr(ws), C.c_size_t(ws.numel()), cur_stream()), "conv2d")
    return y


def conv2d_gn(x, w_krsc, gamma, beta, bias=None, res=None, stride=1, upsample2x=False, groups=32, eps=1e-5, swish=False):
    """[conv2d whose epilogue emits the GroupNorm partials of its output] -> [normalise-only GroupNorm]: returns (y, GroupNorm(y), slots)
    or None when the plan of this shape cannot emit partials (split-K, strips cutting a group, ...)."""
    lib = _lib.load()
    _need_cuda(x, w_krsc, gamma, beta)
    n, h, w, cin = x.shape
    cout, k, _, cin_w = w_krsc.shape
    assert cin == cin_w and x.is_contiguous() and w_krsc.is_contiguous()
    hv, wv = (2 * h, 2 * w) if upsample2x else (h, w)
    pad = k // 2
    ho = (hv + 2 * pad - k) // stride + 1
    wo = (wv + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x.device)
    yn = torch.empty_like(y)
    pf = n * (ho * wo) * groups * 2 // 16 + n * groups * 2 + 1024           # tiles hold >= 32 rows: more than any plan needs
    part = torch.empty(pf, dtype=torch.float32, device=x.device)
    slots = C.c_int(0)
    check(lib.sdeo_debug_conv2d_gn_f16(ptr(yn), ptr(y), ptr(x), ptr(w_krsc), ptr(bias), ptr(res), _i(n), _i(h), _i(w), _i(cin), _i(cout),
                                       _i(k), _i(stride), _i(int(upsample2x)), ptr(gamma), ptr(beta), _i(groups), _f(eps), _i(int(swish)),
                                       ptr(part), C.c_size_t(pf), C.byref(slots), cur_stream()), "conv2d_gn")
    if slots.value == 0:
        return None
    return y, yn, slots.value


def conv3x3_gn_in(x, w_krsc, gamma, beta, partials, bias=None, eps=1e-5, swish=True):
    """conv3x3(silu(GroupNorm32(x))) with the normalisation applied inside the conv kernel from `partials` [n][slots][32][2] = partial
    (sum, sumsq) of x per (image, slot, group); None when the plan of this shape cannot."""
    lib = _lib.load()
    _need_cuda(x, w_krsc, gamma, beta, partials)
    n, h, w, cin = x.shape
    cout = w_krsc.shape[0]
    assert x.is_contiguous() and w_krsc.is_contiguous() and partials.dtype == torch.float32 and partials.is_contiguous()
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device)
    ws = _ws(64 << 20, x.device)
    ok = C.c_int(0)
    check(lib.sdeo_debug_conv2d_gnin_f16(ptr(y), ptr(x), ptr(w_krsc), ptr(bias), _i(n), _i(h), _i(w), _i(cin), _i(cout), ptr(gamma), ptr(beta),
                                         ptr(partials), _i(partials.shape[1]), _f(eps), _i(int(swish)), ptr(ws), C.c_size_t(ws.numel()),
                                         C.byref(ok), cur_stream()), "conv3x3_gn_in")
    return y if ok.value else None


def gemm(x, w, bias=None, res=None, act=0, scale=1.0, out_f32=False, bias_per_row=False, w8=None):
    """y[m][n] = x[m][k] . w[n][k]^T (+bias)(+res); x, w fp16 row-major (may be strided views with unit inner stride).
    w8 = (codes, scales): stream the fp8 copy of w instead."""
    lib = _lib.load()
    _need_cuda(x, w)
    m, k = x.shape
    n, k2 = w.shape
    assert k == k2 and x.stride(1) == 1 and w.stride(1) == 1
    y = torch.empty((m, n), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    nb = lib.sdeo_gemm_workspace_bytes(_i(m), _i(n), _i(k))
    ws = _ws(nb if w8 is None else max(nb, 64 << 20), x.device)
    _arm_fp8(lib, w8)
    check(lib.sdeo_gemm_f16(ptr(y), _i(n), ptr(x), _i(x.stride(0)), ptr(w), _i(w.stride(0)), ptr(bias), ptr(res),
                            _i(res.stride(0) if res is not None else 0), _i(m), _i(n), _i(k), _i(act), _f(scale),
                            _i(int(out_f32)), _i(int(bias_per_row)), ptr(ws), C.c_size_t(ws.numel()), cur_stream()), "gemm")
    return y


def fold_layernorm(w, gamma, beta, bias=None):
    """(w * gamma as fp16 [rows][c], row sums s fp32 [rows], bias + w beta fp32 [rows]): the load-time LayerNorm fold."""
    lib = _lib.load()
    _need_cuda(w, gamma, beta)
    rows, c = w.shape
    assert w.dtype == torch.float16 and w.is_contiguous()
    wo = torch.empty_like(w)
    s = torch.empty((rows,), dtype=torch.float32, device=w.device)
    b = torch.empty((rows,), dtype=torch.float32, device=w.device)
    check(lib.sdeo_debug_fold_layernorm(ptr(wo), ptr(s), ptr(b), ptr(w), ptr(gamma), ptr(beta), ptr(bias), _i(rows), _i(c),
                                        cur_stream()), "fold_layernorm")
    return wo, s, b


def compose_proj(wp, bp, w2, b2):
    """([ (wp w2) | wp ] as fp16 [c][k2 + c], wp b2 + bp fp32 [c]): ff.net.2 followed by proj_out as one Linear over [g | t]."""
    lib = _lib.load()
    _need_cuda(wp, bp, w2, b2)
    c, k2 = w2.shape
    assert wp.shape == (c, c) and wp.dtype == torch.float16 and w2.dtype == torch.float16 and wp.is_contiguous() and w2.is_contiguous()
    wo = torch.empty((c, k2 + c), dtype=torch.float16, device=wp.device)
    bo = torch.empty((c,), dtype=torch.float32, device=wp.device)
    check(lib.sdeo_debug_compose_proj(ptr(wo), ptr(bo), ptr(wp), ptr(bp), ptr(w2), ptr(b2), _i(c), _i(k2), cur_stream()), "compose_proj")
    return wo, bo


def gemm_with_row_stats(x, w, bias=None, res=None):
    """y = x w^T (+bias)(+res) in fp16 plus the per-row (sum, sumsq) partials its epilogue writes: (y, stats [m][ld][2], strips).
    When the plan for this shape is split-K the statistics come from the row_stats kernel (strips = 1), as in the networks."""
    lib = _lib.load()
    _need_cuda(x, w)
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float16, device=x.device)
    ld = max(1, (n + 31) // 32)
    stats = torch.zeros((m, ld, 2), dtype=torch.float32, device=x.device)
    strips = C.c_int(0)
    check(lib.sdeo_debug_gemm_stats_f16(ptr(y), _i(n), ptr(x), _i(x.stride(0)), ptr(w), _i(w.stride(0)), ptr(bias), ptr(res),
                                        _i(res.stride(0) if res is not None else 0), _i(m), _i(n), _i(k), ptr(stats), _i(ld),
                                        C.byref(strips), cur_stream()), "gemm_stats")
    if strips.value == 0:
        y = gemm(x, w, bias=bias, res=res)
        check(lib.sdeo_debug_row_stats_f16(ptr(stats), _i(ld), ptr(y), _i(n), _i(m), _i(n), cur_stream()), "row_stats")
        return y, stats, 1
    return y, stats, strips.value


def gemm_layernorm(x, stats, strips, w_folded, ln_s, bias_folded, act=0, eps=1e-5):
    """y[m][n] = act(LN(x)[m] . w[n] + b[n]) with x [m][k] raw and the fold of (w, gamma, beta, bias)."""
    lib = _lib.load()
    _need_cuda(x, w_folded, stats)
    m, k = x.shape
    n = w_folded.shape[0]
    y = torch.empty((m, n // 2 if act == 3 else n), dtype=torch.float16, device=x.device)
    ws = _ws(lib.sdeo_gemm_workspace_bytes(_i(m), _i(n), _i(k)), x.device)
    check(lib.sdeo_debug_gemm_ln_f16(ptr(y), _i(y.shape[1]), ptr(x), _i(x.stride(0)), ptr(w_folded), _i(w_folded.stride(0)),
                                     ptr(ln_s), ptr(bias_folded), ptr(stats), _i(stats.shape[1]), _i(strips), _i(k),
                                     _i(m), _i(n), _i(k), _i(act), _f(eps), ptr(ws), C.c_size_t(ws.numel()), cur_stream()),
          "gemm_ln")
    return y


def geglu_interleave(w):
    """Row order the GEGLU-fused projection expects (same map as the load-time kernel `geglu_interleave_kernel`):
    [2H][...] with rows 0..H-1 = values, H..2H-1 = gates -> blocks of 16 alternating value / gate."""
    h = w.shape[0] // 2
    assert h % 16 == 0
    v = w[:h].reshape(h // 16, 1, 16, *w.shape[1:])
    g = w[h:].reshape(h // 16, 1, 16, *w.shape[1:])
    return torch.cat([v, g], 1).reshape(w.shape).contiguous()


def gemm_geglu(x, w_interleaved, bias_interleaved=None):
    """y[m][0:H] = (x w_v^T + b_v) * gelu_erf(x w_g^T + b_g): ff.net.0.proj + GEGLU (`attention.py:49-56`) in one launch.
    Weights / bias already in `geglu_interleave` order."""
    lib = _lib.load()
    _need_cuda(x, w_interleaved)
    m, k = x.shape
    n, k2 = w_interleaved.shape
    assert k == k2 and n % 32 == 0 and x.stride(1) == 1 and w_interleaved.stride(1) == 1
    y = torch.empty((m, n // 2), dtype=torch.float16, device=x.device)
    check(lib.sdeo_gemm_f16(ptr(y), _i(n // 2), ptr(x), _i(x.stride(0)), ptr(w_interleaved), _i(w_interleaved.stride(0)),
                            ptr(bias_interleaved), None, _i(0), _i(m), _i(n), _i(k), _i(3), _f(1.0), _i(0), _i(0), None,
                            C.c_size_t(0), cur_stream()), "gemm_geglu")
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    _need_cuda(x, gamma, beta)
    rows, c = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    check(lib.sdeo_layernorm_f16(ptr(y), ptr(x), ptr(gamma), ptr(beta), _i(rows), _i(c), _f(eps), cur_stream()), "layernorm")
    return y


def attention(q, k, v, heads, tk=None, scale=None, causal=False):
    """q (B,Tq,H*d), k (B,TkS,H*d), v (B,TkS,H*d) fp16 -> (B,Tq,H*d); causal masks key j > query t (Tq == tk)."""
    lib = _lib.load()
    _need_cuda(q, k, v)
    b, tq, c = q.shape
    tks = k.shape[1]
    tk = tks if tk is None else tk
    d = c // heads
    scale = d ** -0.5 if scale is None else scale
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and v.shape == k.shape
    o = torch.empty_like(q)
    fn = lib.sdeo_attention_causal_f16 if causal else lib.sdeo_attention_f16
    check(fn(ptr(o), _i(c), ptr(q), _i(c), ptr(k), _i(k.shape[2]), ptr(v), _i(v.shape[2]), _i(b), _i(heads),
             _i(tq), _i(tk), _i(tks), _i(tks), _i(d), _f(scale), cur_stream()), "attention")
    return o


def geglu(a):
    lib = _lib.load()
    _need_cuda(a)
    rows, c2 = a.shape
    y = torch.empty((rows, c2 // 2), dtype=torch.float16, device=a.device)
    check(lib.sdeo_geglu_f16(ptr(y), ptr(a), _i(rows), _i(c2 // 2), cur_stream()), "geglu")
    return y


def timestep_embedding(t, dim):
    lib = _lib.load()
    _need_cuda(t)
    assert t.dtype == torch.int64
    out = torch.empty((t.shape[0], dim), dtype=torch.float16, device=t.device)
    check(lib.sdeo_timestep_embedding_f16(ptr(out), ptr(t), _i(t.shape[0]), _i(dim), cur_stream()), "timestep_embedding")
    return out


def cfg_ddim_step(x, eps_c, eps_u, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None, want_pred_x0=True):
    lib = _lib.load()
    _need_cuda(x, eps_c)
    assert x.dtype == torch.float32 and x.is_contiguous() and eps_c.is_contiguous()
    x_prev = torch.empty_like(x)
    p0 = torch.empty_like(x) if want_pred_x0 else None
    check(lib.sdeo_cfg_ddim_step(ptr(x_prev), ptr(p0), ptr(x), ptr(eps_c), ptr(eps_u), ptr(noise), _f(cfg_scale), _f(a_t),
                                 _f(a_prev), _f(sigma_t), _f(sqrt_one_minus_at), C.c_int64(x.numel()), cur_stream()),
          "cfg_ddim_step")
    return x_prev, p0


def nchw_to_nhwc_f16(x, c_pad=None):
    lib = _lib.load()
    _need_cuda(x)
    n, c, h, w = x.shape
    cp = c_pad or c
    y = torch.empty((n, h, w, cp), dtype=torch.float16, device=x.device)
    check(lib.sdeo_nchw_f32_to_nhwc_f16(ptr(y), _i(cp), ptr(x.contiguous().float()), _i(n), _i(c), _i(h * w), cur_stream()),
          "nchw_to_nhwc")
    return y


def nhwc_to_nchw_f32(x, c=None, scale=1.0):
    lib = _lib.load()
    _need_cuda(x)
    n, h, w, ld = x.shape
    c = c or ld
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib.sdeo_nhwc_f16_to_nchw_f32(ptr(y), ptr(x), _i(ld), _i(n), _i(c), _i(h * w), _f(scale), cur_stream()),
          "nhwc_to_nchw")
    return y
