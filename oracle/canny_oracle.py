"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of `cv2.Canny(img, low, high)` as the reference calls it
(`annotator/canny/__init__.py:4-6`: default apertureSize = 3, L2gradient = False, on the 3-channel uint8 image produced by
`resize_image(HWC3(input_image), ...)`, `canny2image_torch.py:30-33`).

The arithmetic lives in a third-party dependency that is absent from /root/reference and from this image: OpenCV, pinned by the
reference at opencv-contrib-python == 4.3.0.36 (`environment.yaml:15`).  This file restates the published algorithm of
modules/imgproc/src/canny.cpp of that release (integer path):

  1. dx, dy = Sobel(src, CV_16S, aperture 3, BORDER_REPLICATE) per channel;
  2. magnitude |dx| + |dy| (L1); with several channels every pixel keeps the channel of the largest magnitude (first one wins a tie);
  3. non-maximum suppression on pixels with mag > low, direction by the fixed-point sector test
     (TG22 = round(tan(22.5 deg) * 2^15) = 13573): horizontal  m > left  && m >= right;  vertical  m > up && m >= down;
     diagonal  m > and m > (both strict) along the sign of dx*dy; magnitudes outside the image are 0;
  4. survivors with mag > high are edges, the others are candidates; candidates 8-connected to an edge become edges (hysteresis);
  5. output 255 on edges, 0 elsewhere.  low / high are floored and swapped when low > high.

PARITY UNPINNED: no cv2 in the container and the reference holds no stored edge map, so this restatement is checked only against
hand-computable cases (tests/test_canny_oracle.py) and frozen as a regression fixture (tests/golden/canny.npz); the HIP kernel is
held bit-exact to it."""
from __future__ import annotations

import numpy as np

TG22 = 13573


def sobel3(ch: np.ndarray):
    """int32 dx, dy of one uint8 channel, 3x3 Sobel, replicated border."""
    p = np.pad(ch.astype(np.int32), 1, mode="edge")
    a, b, c = p[:-2, :-2], p[:-2, 1:-1], p[:-2, 2:]
    d, f = p[1:-1, :-2], p[1:-1, 2:]
    g, h, i = p[2:, :-2], p[2:, 1:-1], p[2:, 2:]
    dx = (c + 2 * f + i) - (a + 2 * d + g)
    dy = (g + 2 * h + i) - (a + 2 * b + c)
    return dx, dy


def gradient(img: np.ndarray):
    """per-pixel (dx, dy, mag) of the channel with the largest L1 magnitude (first channel wins ties)."""
    if img.ndim == 2:
        img = img[:, :, None]
    assert img.dtype == np.uint8
    H, W, C = img.shape
    best = None
    for k in range(C):
        dx, dy = sobel3(img[:, :, k])
        mag = np.abs(dx) + np.abs(dy)
        if best is None:
            best = [dx, dy, mag]
        else:
            take = mag > best[2]
            best = [np.where(take, dx, best[0]), np.where(take, dy, best[1]), np.where(take, mag, best[2])]
    return best


def nms_map(dx, dy, mag, low: int, high: int):
    """0 = candidate (kept, <= high), 1 = not an edge, 2 = edge (kept, > high)."""
    H, W = mag.shape
    m = np.pad(mag, 1, mode="constant")              # zeros outside the image
    c = m[1:-1, 1:-1]
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = ~(horiz | vert)
    s = np.where((dx ^ dy) < 0, -1, 1)
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    prev_d = m[ii, jj - s + 1]                        # row above, column j - s   (padded coordinates: row ii+1-1, col jj+1-s)
    next_d = m[ii + 2, jj + s + 1]                    # row below, column j + s
    keep = (horiz & (c > left) & (c >= right)) | (vert & (c > up) & (c >= down)) | (diag & (c > prev_d) & (c > next_d))
    keep &= c > low
    out = np.ones((H, W), np.uint8)
    out[keep & (c > high)] = 2
    out[keep & (c <= high)] = 0
    return out


def hysteresis(pmap: np.ndarray):
    """edges = 8-connected components of (candidate | edge) that contain an edge."""
    from scipy import ndimage
    lab, n = ndimage.label(pmap != 1, structure=np.ones((3, 3), np.int32))
    strong = np.zeros(n + 1, bool)
    strong[np.unique(lab[pmap == 2])] = True
    strong[0] = False
    return strong[lab]


def canny(img: np.ndarray, low_threshold, high_threshold) -> np.ndarray:
    low, high = int(np.floor(low_threshold)), int(np.floor(high_threshold))
    if low > high:
        low, high = high, low
    dx, dy, mag = gradient(img)
    return (hysteresis(nms_map(dx, dy, mag, low, high)) * 255).astype(np.uint8)


def hwc3(x: np.ndarray) -> np.ndarray:
    """`annotator/util.py:9-25`."""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    H, W, C = x.shape
    assert C in (1, 3, 4)
    if C == 3:
        return x
    if C == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    return (color * alpha + 255.0 * (1.0 - alpha)).clip(0, 255).astype(np.uint8)


def control_from_edges(edges: np.ndarray, num_samples: int) -> np.ndarray:
    """`canny2image_torch.py:34-38`: HWC3(detected_map) / 255 -> (B, 3, H, W) float32."""
    c = hwc3(edges).astype(np.float32) / 255.0
    return np.stack([c] * num_samples).transpose(0, 3, 1, 2).copy()
