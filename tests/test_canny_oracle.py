"""CPU checks of oracle/canny_oracle.py (the numpy restatement of cv2.Canny the HIP kernel is held to).  PARITY UNPINNED against
OpenCV itself (not importable here, no stored edge map in the reference): the cases below are hand-computable from the published
algorithm, plus the frozen regression fixture tests/golden/canny.npz (input = a reference test picture)."""
import os

import numpy as np

from oracle import canny_oracle as O
from tests.common import GOLDEN


def test_constant_image_has_no_edges():
    img = np.full((9, 13, 3), 77, np.uint8)
    assert O.canny(img, 100, 200).max() == 0


def test_vertical_step_gives_one_column():
    # columns 0..7 = 0, 8..15 = 255: dx = 4*255 at x = 7 and x = 8; NMS (m > left && m >= right) keeps x = 7 only
    img = np.zeros((8, 16), np.uint8)
    img[:, 8:] = 255
    e = O.canny(img, 100, 200)
    want = np.zeros_like(e)
    want[:, 7] = 255
    np.testing.assert_array_equal(e, want)


def test_horizontal_step_gives_one_row():
    img = np.zeros((12, 10, 3), np.uint8)
    img[6:] = 255
    e = O.canny(img, 100, 200)
    want = np.zeros((12, 10), np.uint8)
    want[5] = 255                       # m > up && m >= down keeps the upper of the two rows
    np.testing.assert_array_equal(e, want)


def test_sobel_replicated_border_and_channel_choice():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (5, 6, 3), dtype=np.uint8)
    dx, dy, mag = O.gradient(img)
    # pixel (0, 0), channel k: replicated border => rows -1 = 0, cols -1 = 0
    best = None
    for k in range(3):
        c = img[:, :, k].astype(int)
        p = lambda y, x: c[min(max(y, 0), 4), min(max(x, 0), 5)]
        gx = (p(-1, 1) + 2 * p(0, 1) + p(1, 1)) - (p(-1, -1) + 2 * p(0, -1) + p(1, -1))
        gy = (p(1, -1) + 2 * p(1, 0) + p(1, 1)) - (p(-1, -1) + 2 * p(-1, 0) + p(-1, 1))
        m = abs(gx) + abs(gy)
        if best is None or m > best[2]:
            best = (gx, gy, m)
    assert (dx[0, 0], dy[0, 0], mag[0, 0]) == best


def test_hysteresis_keeps_only_candidates_connected_to_an_edge():
    pmap = np.ones((6, 12), np.uint8)
    pmap[1, 1:6] = 0
    pmap[1, 6] = 2                      # chain touching an edge (8-connected through the diagonal below)
    pmap[2, 7] = 0
    pmap[4, 1:5] = 0                    # isolated chain of candidates
    e = O.hysteresis(pmap)
    assert e[1, 1:7].all() and e[2, 7] and not e[4].any() and e.sum() == 7


def test_thresholds_are_floored_and_swapped():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    a = O.canny(img, 100, 200)
    np.testing.assert_array_equal(a, O.canny(img, 200, 100))
    np.testing.assert_array_equal(a, O.canny(img, 100.9, 200.9))


def test_regression_fixture():
    g = np.load(os.path.join(GOLDEN, "canny.npz"))
    e = O.canny(g["image"], int(g["low"]), int(g["high"]))
    want = np.unpackbits(g["edges"])[:e.size].reshape(e.shape) * 255
    np.testing.assert_array_equal(e, want.astype(np.uint8))


def test_control_from_edges():
    e = np.zeros((4, 6), np.uint8)
    e[1, 2] = 255
    c = O.control_from_edges(e, 2)
    assert c.shape == (2, 3, 4, 6) and c.dtype == np.float32 and c[1, 2, 1, 2] == 1.0 and c.sum() == 6.0
