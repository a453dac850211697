"""`annotator/util.py` of the reference: `HWC3` and `resize_image` (same names, same semantics), with `cv2.resize` on the GPU.

`resize_image` (`annotator/util.py:28-38`) scales the shorter side to `resolution`, rounds both sides to multiples of 64 and calls
`cv2.resize(..., INTER_LANCZOS4 if k > 1 else INTER_AREA)`.  Here the pixels are resampled by csrc/resize.hip
(`sdeo_resize_lanczos4_u8` / `sdeo_resize_area_u8`); this module builds the per-axis coefficient tables those kernels gather
through -- O(W + H) numbers that depend only on the sizes (OpenCV 4.3 `resize.cpp`: `interpolateLanczos4`,
`computeResizeAreaTab`).  numpy in -> numpy out (the reference's contract); CUDA tensor in -> CUDA tensor out, so `process()`
keeps the image on the device from here through Canny to the control tensor.  There is no host resampling path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import check, cur_stream, ptr


def HWC3(x):
    """`annotator/util.py:9-25`."""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    H, W, C_ = x.shape
    assert C_ in (1, 3, 4)
    if C_ == 3:
        return x
    if C_ == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    return (color * alpha + 255.0 * (1.0 - alpha)).clip(0, 255).astype(np.uint8)


def target_size(H, W, resolution):
    """size rule of `annotator/util.py:28-36` (shorter side -> resolution, both rounded to multiples of 64)."""
    k = float(resolution) / min(H, W)
    return int(np.round(H * k / 64.0)) * 64, int(np.round(W * k / 64.0)) * 64


# ---------------------------------------------------------------------------------------------- coefficient tables
_S45 = 0.70710678118654752440084436210485
_CS = np.array([[1, 0], [-_S45, -_S45], [0, 1], [_S45, -_S45], [-1, 0], [_S45, _S45], [0, -1], [-_S45, _S45]], dtype=np.float64)


def _lanczos4_axis(ssize: int, dsize: int):
    """first tap (source index of tap 0) and the 8 short coefficients (x 2048) of every destination index"""
    scale = np.float64(ssize) / np.float64(dsize)
    f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    x = (f - s).astype(np.float32)
    co = np.zeros((dsize, 8), dtype=np.float32)
    tiny = x < np.finfo(np.float32).eps
    co[tiny, 3] = 1.0
    xs = x[~tiny].astype(np.float64)
    y0 = -(xs + 3) * np.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    c = np.zeros((xs.shape[0], 8), dtype=np.float32)
    for i in range(8):
        y = -(xs + 3 - i) * np.pi * 0.25
        c[:, i] = ((_CS[i, 0] * s0 + _CS[i, 1] * c0) / (y * y)).astype(np.float32)
    tot = np.zeros(xs.shape[0], dtype=np.float32)
    for i in range(8):
        tot = (tot + c[:, i]).astype(np.float32)
    co[~tiny] = (c * (np.float32(1.0) / tot).astype(np.float32)[:, None]).astype(np.float32)
    return (s - 3).astype(np.int32), np.clip(np.rint(co * np.float32(2048.0)), -32768, 32767).astype(np.int16)


def _area_axis(ssize: int, dsize: int):
    """CSR table (start[dsize + 1], index, weight) of INTER_AREA: covered fractions when the axis shrinks, OpenCV's area-mode
    linear pair when it grows.

    Known deviations from cv2 (cv2 is not installed in this image, so INTER_AREA is "parity unpinned"; the Lanczos path, the one
    `process()` takes for the benchmark sizes, is the integer-exact one):
      * (both axes shrinking by integer factors do not come here: `resize_u8` takes cv2's `resizeAreaFast_` arithmetic,
        `sdeo_resize_area_fast_u8`)
      * an axis that GROWS under INTER_AREA: cv2 runs its u8 bilinear kernel in 11-bit fixed point (weights x 2048, two rounding
        shifts); here the linear pair is applied in fp32 -- differences of at most 1 code."""
    scale = np.float64(ssize) / np.float64(dsize)
    start, idx, wgt = [0], [], []
    for d in range(dsize):
        if ssize >= dsize:
            f1 = d * scale
            f2 = f1 + scale
            cell = min(scale, ssize - f1)
            s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
            s2 = min(s2, ssize - 1)
            s1 = min(s1, s2)
            if s1 - f1 > 1e-3:
                idx.append(s1 - 1); wgt.append(np.float32((s1 - f1) / cell))
            for sx in range(s1, s2):
                idx.append(sx); wgt.append(np.float32(1.0 / cell))
            if f2 - s2 > 1e-3:
                idx.append(s2); wgt.append(np.float32(min(min(f2 - s2, 1.0), cell) / cell))
        else:
            sx = int(np.floor(d * scale))
            fx = np.float32((d + 1) - (sx + 1) * (1.0 / scale))
            fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.floor(fx))
            if sx < 0:
                sx, fx = 0, np.float32(0.0)
            if sx >= ssize - 1:
                sx, fx = ssize - 1, np.float32(0.0)
            idx.append(sx); wgt.append(np.float32(1.0) - fx)
            if fx > 0 and sx + 1 < ssize:
                idx.append(sx + 1); wgt.append(fx)
        start.append(len(idx))
    return np.asarray(start, dtype=np.int32), np.asarray(idx, dtype=np.int32), np.asarray(wgt, dtype=np.float32)


# ---------------------------------------------------------------------------------------------- cv2.resize on the device
def resize_u8(img, dst_h: int, dst_w: int, interpolation: str):
    """`cv2.resize(img, (dst_w, dst_h), interpolation)` for an HWC uint8 image; interpolation 'lanczos4' | 'area'.
    numpy in -> numpy out, CUDA tensor in -> CUDA tensor out."""
    if not torch.cuda.is_available():
        raise _lib.SdeoError("resize_image needs a HIP device (there is no CPU fallback)")
    lib = _lib.load()
    as_np = isinstance(img, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(img)) if as_np else img
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise _lib.SdeoError(f"resize: HWC uint8 image expected, got {t.dtype} with {t.dim()} dims")
    h, w, c = t.shape
    if (h, w) == (dst_h, dst_w):
        return img.copy() if as_np else img.clone()
    t = t.to("cuda").contiguous()
    out = torch.empty((dst_h, dst_w, c), dtype=torch.uint8, device=t.device)
    dev = lambda a: torch.from_numpy(a).to(t.device)
    geo = (C.c_int(h), C.c_int(w), C.c_int(c), C.c_int(dst_h), C.c_int(dst_w))
    if interpolation == "lanczos4":
        x0, ax = _lanczos4_axis(w, dst_w)
        y0, by = _lanczos4_axis(h, dst_h)
        tabs = [dev(a) for a in (x0, ax, y0, by)]
        check(lib.sdeo_resize_lanczos4_u8(ptr(out), ptr(t), *geo, *[ptr(a) for a in tabs], cur_stream()), "resize_lanczos4")
    elif interpolation == "area" and h % dst_h == 0 and w % dst_w == 0:
        # both axes shrink by integer factors: cv2 switches to resizeAreaFast_ (integer cell sums), so does this
        check(lib.sdeo_resize_area_fast_u8(ptr(out), ptr(t), *geo, cur_stream()), "resize_area_fast")
    elif interpolation == "area":
        tabs = [dev(a) for a in (*_area_axis(w, dst_w), *_area_axis(h, dst_h))]
        check(lib.sdeo_resize_area_u8(ptr(out), ptr(t), *geo, *[ptr(a) for a in tabs], cur_stream()), "resize_area")
    else:
        raise ValueError(f"interpolation {interpolation!r}: 'lanczos4' or 'area'")
    if as_np:
        return out.cpu().numpy()
    torch.cuda.current_stream().synchronize()      # the tables are freed with this frame: make sure the kernel is done with them
    return out


def resize_image(input_image, resolution):
    """`annotator/util.py:28-38`."""
    H, W, _ = input_image.shape
    k = float(resolution) / min(float(H), float(W))
    Ht, Wt = target_size(H, W, resolution)
    return resize_u8(input_image, Ht, Wt, "lanczos4" if k > 1 else "area")
