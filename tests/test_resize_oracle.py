"""oracle/resize_oracle.py (the restatement of cv2.resize INTER_LANCZOS4 / INTER_AREA for 8-bit images) against properties that
follow from OpenCV's definition of the two interpolations.  PARITY UNPINNED against OpenCV itself: cv2 is not importable here
and the reference stores no resized image (stated in the oracle's header and in DESIGN.md)."""
import numpy as np

from oracle import resize_oracle as R


def rnd(h, w, c=3, seed=0):
    return (np.random.RandomState(seed).rand(h, w, c) * 255).astype(np.uint8)


def test_identity_and_size_rule():
    img = rnd(256, 384)
    assert np.array_equal(R.resize_image(img, 256), img)          # the reference's own data: 384x256 at resolution 256
    assert R.resize_image(rnd(100, 150), 256).shape == (256, 384, 3)
    assert R.resize_image(rnd(500, 333), 512).shape == (768, 512, 3)
    assert R.resize_image(rnd(300, 700), 128).shape == (128, 320, 3)


def test_area_integer_factor_is_the_block_mean():
    img = rnd(64, 96, 3, 1)
    out = R.cv2_resize(img, (48, 32), "area")                      # 2 x 2 blocks
    mean = img.reshape(32, 2, 48, 2, 3).astype(np.float64).mean((1, 3))
    assert np.abs(out.astype(np.float64) - mean).max() <= 0.5 + 1e-9
    out3 = R.cv2_resize(img, (32, 16), "area")                     # 4 x 3 blocks (rows x cols: 64/16 = 4, 96/32 = 3)
    mean3 = img.reshape(16, 4, 32, 3, 3).astype(np.float64).mean((1, 3))
    assert np.abs(out3.astype(np.float64) - mean3).max() <= 0.5 + 1e-4


def test_area_fast_path_arithmetic():
    """Both axes shrinking by integer factors: OpenCV's `resizeAreaFast_` arithmetic (known answers worked by hand).
    2 x 2: (a + b + c + d + 2) >> 2, i.e. ties round UP (sum 10 -> 3, where round-half-even of 2.5 would give 2); other cells:
    int sum * float32(1 / area), ties to even (a 2 x 1 cell with sum 5 -> 2, with sum 7 -> 4)."""
    img = np.array([[[1], [2], [0], [0]], [[3], [4], [1], [1]]], dtype=np.uint8)         # cells: sum 10 -> 3, sum 2 -> 1
    assert R.cv2_resize(img, (2, 1), "area").ravel().tolist() == [3, 1]
    col = np.array([[[2]], [[3]], [[3]], [[4]]], dtype=np.uint8)                           # 2 x 1 cells: sums 5 and 7
    assert R.cv2_resize(col, (1, 2), "area").ravel().tolist() == [2, 4]
    big = rnd(48, 60, 3, 5)
    out = R.cv2_resize(big, (20, 12), "area")                                              # 4 x 3 cells
    cells = big.astype(np.int64).reshape(12, 4, 20, 3, 3).sum((1, 3))
    assert np.array_equal(out, np.rint(cells.astype(np.float32) * (np.float32(1.0) / np.float32(12.0))).astype(np.uint8))


def test_area_fractional_weights_sum_to_one_and_cover_the_cell():
    for s, d in ((100, 64), (333, 128), (97, 96), (1000, 7)):
        for ent in R.area_tab(s, d):
            assert abs(sum(float(w) for _, w in ent) - 1.0) < 1e-5
            idx = [i for i, _ in ent]
            assert idx == list(range(idx[0], idx[0] + len(idx))) and 0 <= idx[0] and idx[-1] < s


def test_lanczos_coefficients():
    co = R.lanczos4_coeffs(np.array([0.0, 0.5, 0.25], dtype=np.float32))
    assert np.array_equal(co[0], np.array([0, 0, 0, 1, 0, 0, 0, 0], dtype=np.float32))
    assert np.allclose(co.sum(1), 1.0, atol=1e-6)
    assert np.allclose(co[1], co[1][::-1], atol=1e-6)              # x = 0.5 is symmetric
    # closed form vs the textbook Lanczos-4 window  sinc(t) sinc(t / 4)
    t = np.arange(-3, 5) - 0.25
    ref = np.sinc(t) * np.sinc(t / 4)
    assert np.allclose(co[2], ref / ref.sum(), atol=2e-6)


def test_lanczos_constant_image_and_impulse():
    assert np.array_equal(R.cv2_resize(np.full((20, 30, 3), 77, np.uint8), (45, 40), "lanczos4"), np.full((40, 45, 3), 77, np.uint8))
    img = np.zeros((32, 32, 1), np.uint8)
    img[16, 16] = 255
    out = R.cv2_resize(img, (64, 64), "lanczos4").astype(np.int32)[:, :, 0]
    y0, by = R.lanczos4_tab(32, 64)
    x0, ax = R.lanczos4_tab(32, 64)
    dy, dx = 33, 30
    ky, kx = 16 - y0[dy], 16 - x0[dx]
    want = (255 * int(ax[dx, kx]) * int(by[dy, ky]) + (1 << 21)) >> 22
    assert out[dy, dx] == min(max(want, 0), 255)


def test_area_upscaled_axis_uses_two_taps():
    img = rnd(64, 60, 3, 3)
    out = R.cv2_resize(img, (64, 32), "area")                      # height halves, width grows 60 -> 64
    assert out.shape == (32, 64, 3)
    t = R.area_linear_tab(60, 64)
    assert all(1 <= len(e) <= 2 and abs(sum(float(w) for _, w in e) - 1.0) < 1e-6 for e in t)
