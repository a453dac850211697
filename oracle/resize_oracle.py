"""CPU restatement of `cv2.resize(img, (W, H), interpolation=INTER_LANCZOS4 | INTER_AREA)` for 8-bit images, the call
`annotator/util.py:28-38` (`resize_image`) makes.  TEST INFRASTRUCTURE ONLY.

The algorithm lives in a third-party dependency that is absent from /root/reference and from this image: OpenCV
(`opencv-python==4.3.0.36`, `requirements.txt`), `modules/imgproc/src/resize.cpp`.  Restated from the published source:

INTER_LANCZOS4 (`resizeGeneric_` with `HResizeLanczos4<uchar, int, short>` / `VResizeLanczos4<..., FixedPtCast<int, uchar, 22>>`):
    fx = (dx + 0.5) * scale_x - 0.5;  sx = floor(fx);  fx -= sx;  taps sx-3 .. sx+4, indices clamped to the image (replicate);
    8 coefficients by `interpolateLanczos4` (closed form on a sin / cos pair in double, normalised in float), stored as
    short = cvRound(coeff * 2048);  horizontal and vertical passes in 32-bit integers;  out = saturate((sum + 2^21) >> 22).
INTER_AREA with both scales >= 1 (`resizeArea_<uchar, float>` on the `computeResizeAreaTab` tables):
    a destination cell covers [d * scale, (d + 1) * scale) of the source; weights = covered fraction / cell width, accumulated
    in float (no fused multiply-add), rows then columns in the table order; out = saturate(cvRound(sum)) (ties to even).
    (OpenCV takes an integer-only fast path when both scales are integers; its result is the same block mean up to the rounding
    of one float division, which this restatement does not special-case.)
INTER_AREA with a scale < 1 on either axis is OpenCV's linear interpolation with area-style coefficients; `resize_image`
reaches it only when rounding a side to a multiple of 64 enlarges it although k <= 1, and it is restated here the same way
(`area_linear_tab`).

**Parity unpinned**: cv2 is not importable in this container and the reference stores no resized image, so nothing pins this
restatement to OpenCV's output; hand-computable properties are tested instead (identity, block means, constant images,
impulse responses equal to the coefficient table)."""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def lanczos4_coeffs(x: np.ndarray) -> np.ndarray:
    """`interpolateLanczos4`: x float32 in [0, 1) -> [n][8] float32"""
    x = np.asarray(x, dtype=np.float32)
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]], dtype=np.float64)
    out = np.zeros((x.shape[0], 8), dtype=np.float32)
    tiny = x < np.finfo(np.float32).eps
    out[tiny, 3] = 1.0
    xs = x[~tiny].astype(np.float64)
    y0 = -(xs + 3) * np.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    c = np.zeros((xs.shape[0], 8), dtype=np.float32)
    for i in range(8):
        y = -(xs + 3 - i) * np.pi * 0.25
        c[:, i] = ((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y)).astype(np.float32)
    ssum = np.zeros(xs.shape[0], dtype=np.float32)
    for i in range(8):
        ssum = (ssum + c[:, i]).astype(np.float32)
    inv = (np.float32(1.0) / ssum).astype(np.float32)
    out[~tiny] = (c * inv[:, None]).astype(np.float32)
    return out


def lanczos4_tab(ssize: int, dsize: int):
    """per destination index: first tap (sx - 3, may be negative) and the 8 short coefficients"""
    scale = np.float64(ssize) / np.float64(dsize)
    d = np.arange(dsize)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    frac = (f - s).astype(np.float32)
    co = lanczos4_coeffs(frac)
    ico = np.clip(np.rint(co * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int16)
    return s - 3, ico


def resize_lanczos4(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    h, w, c = img.shape
    y0, by = lanczos4_tab(h, dh)
    x0, ax = lanczos4_tab(w, dw)
    src = img.astype(np.int32)
    rows = np.zeros((h, dw, c), dtype=np.int32)
    with np.errstate(over="ignore"):
        for k in range(8):
            xi = np.clip(x0 + k, 0, w - 1)
            rows = (rows + src[:, xi, :] * ax[:, k].astype(np.int32)[None, :, None]).astype(np.int32)
        acc = np.zeros((dh, dw, c), dtype=np.int32)
        for k in range(8):
            yi = np.clip(y0 + k, 0, h - 1)
            acc = (acc + rows[yi] * by[:, k].astype(np.int32)[:, None, None]).astype(np.int32)
        out = (acc + np.int32(1 << (2 * COEF_BITS - 1))) >> np.int32(2 * COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def area_tab(ssize: int, dsize: int):
    """`computeResizeAreaTab`: list per destination index of (source index, float32 weight), scale >= 1"""
    scale = np.float64(ssize) / np.float64(dsize)
    tab = []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def area_linear_tab(ssize: int, dsize: int):
    """INTER_AREA on an axis that grows (scale < 1): OpenCV's area-mode linear coefficients, two taps per destination index"""
    scale = np.float64(ssize) / np.float64(dsize)
    inv = 1.0 / scale
    tab = []
    for d in range(dsize):
        sx = int(np.floor(d * scale))
        fx = np.float32((d + 1) - (sx + 1) * inv)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx < 0:
            sx, fx = 0, np.float32(0.0)
        if sx >= ssize - 1:
            sx, fx = ssize - 1, np.float32(0.0)
        ent = [(sx, np.float32(1.0) - fx)]
        if fx > 0 and sx + 1 < ssize:
            ent.append((sx + 1, fx))
        tab.append(ent)
    return tab


def resize_area_fast(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    """INTER_AREA with both axes shrinking by integer factors: OpenCV 4.x resize.cpp leaves `resizeArea_` for `resizeAreaFast_`
    (`is_area_fast`).  2 x 2 cells of 8-bit images: (a + b + c + d + 2) >> 2 (`ResizeAreaFastVec`); any other cell: the integer cell
    sum times float32(1 / area), cvRound (ties to even), saturate (`ResizeAreaFast_Invoker`).  Restated from the published source;
    parity unpinned (cv2 is not installed here and the reference stores no resized image)."""
    h, w, c = img.shape
    sy, sx = h // dh, w // dw
    assert sy * dh == h and sx * dw == w
    cells = img.astype(np.int64).reshape(dh, sy, dw, sx, c).sum(axis=(1, 3))
    if sy == 2 and sx == 2:
        return ((cells + 2) >> 2).astype(np.uint8)
    scale = np.float32(1.0) / np.float32(sx * sy)
    return np.clip(np.rint((cells.astype(np.float32) * scale).astype(np.float32)), 0, 255).astype(np.uint8)


def resize_area(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    h, w, c = img.shape
    if h % dh == 0 and w % dw == 0:
        return resize_area_fast(img, dh, dw)
    xt = area_tab(w, dw) if w >= dw else area_linear_tab(w, dw)
    yt = area_tab(h, dh) if h >= dh else area_linear_tab(h, dh)
    src = img.astype(np.float32)
    out = np.zeros((dh, dw, c), dtype=np.uint8)
    for dy in range(dh):
        ssum = None
        for (sy, beta) in yt[dy]:
            buf = np.zeros((dw, c), dtype=np.float32)
            for dx in range(dw):
                a = np.zeros(c, dtype=np.float32)
                for (sx, alpha) in xt[dx]:
                    a = (a + src[sy, sx] * alpha).astype(np.float32)
                buf[dx] = a
            ssum = (buf * beta).astype(np.float32) if ssum is None else (ssum + (buf * beta).astype(np.float32)).astype(np.float32)
        out[dy] = np.clip(np.rint(ssum), 0, 255).astype(np.uint8)
    return out


def cv2_resize(img: np.ndarray, dsize, interpolation: str) -> np.ndarray:
    """dsize = (W, H) like cv2; interpolation 'lanczos4' | 'area'"""
    dw, dh = dsize
    if (dh, dw) == img.shape[:2]:
        return img.copy()
    return resize_lanczos4(img, dh, dw) if interpolation == "lanczos4" else resize_area(img, dh, dw)


def resize_image(input_image: np.ndarray, resolution: int) -> np.ndarray:
    """`annotator/util.py:28-38`"""
    H, W, _ = input_image.shape
    k = float(resolution) / min(float(H), float(W))
    Ht = int(np.round(H * k / 64.0)) * 64
    Wt = int(np.round(W * k / 64.0)) * 64
    return cv2_resize(input_image, (Wt, Ht), "lanczos4" if k > 1 else "area")
