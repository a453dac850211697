"""GPU `resize_image` (csrc/resize.hip behind `annotator/util.py: resize_image`) bit-exact against
oracle/resize_oracle.py -- integer arithmetic for Lanczos4, unfused float multiply / add in table order for INTER_AREA.
The oracle itself is parity-unpinned against OpenCV (cv2 is absent; tests/test_resize_oracle.py checks its defining properties)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(h, w, c=3, seed=0):
    return (np.random.RandomState(seed).rand(h, w, c) * 255).astype(np.uint8)


@pytest.mark.parametrize("shape,res", [((256, 384, 3), 256), ((256, 384, 3), 512), ((100, 150, 3), 256), ((500, 333, 3), 512),
                                       ((300, 700, 3), 128), ((720, 1280, 3), 512), ((97, 131, 1), 64), ((64, 60, 4), 64)])
def test_resize_image_bit_exact(shape, res):
    from oracle import resize_oracle as R
    from stablediffusioneo_amd.annotator.util import resize_image
    img = rnd(*shape, seed=shape[0])
    want = R.resize_image(img, res)
    got = resize_image(img, res)
    assert isinstance(got, np.ndarray) and got.dtype == np.uint8 and got.shape == want.shape
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() == 0, f"{shape} -> {want.shape}: {int((d > 0).sum())} of {d.size} differ, max {d.max()}"
    t = resize_image(torch.from_numpy(img).cuda(), res)            # device in -> device out
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), want)


@pytest.mark.parametrize("shape,dst", [((512, 640, 3), (256, 320)), ((96, 120, 3), (24, 40)), ((64, 64, 1), (16, 64)), ((30, 42, 4), (10, 6))])
def test_area_integer_factors_take_the_fast_path(shape, dst):
    """both axes shrink by integer factors (2 x 2: the reference's own bag_scribble.png case, 512 x 640 at resolution 256; 4 x 3; 4 x 1;
    3 x 7): `sdeo_resize_area_fast_u8` bit-exact against the oracle's restatement of OpenCV's resizeAreaFast_"""
    from oracle import resize_oracle as R
    from stablediffusioneo_amd.annotator.util import resize_u8
    img = rnd(*shape, seed=shape[1])
    want = R.resize_area_fast(img, *dst)
    assert np.array_equal(R.cv2_resize(img, (dst[1], dst[0]), "area"), want)
    assert np.array_equal(resize_u8(img, dst[0], dst[1], "area"), want)


def test_resize_extremes():
    from oracle import resize_oracle as R
    from stablediffusioneo_amd.annotator.util import resize_u8
    for img in (np.zeros((40, 50, 3), np.uint8), np.full((40, 50, 3), 255, np.uint8), (np.indices((40, 50)).sum(0) % 2 * 255).astype(np.uint8)[:, :, None]):
        for (dh, dw, interp) in ((96, 128, "lanczos4"), (17, 23, "area"), (20, 64, "area")):
            assert np.array_equal(resize_u8(img, dh, dw, interp), R.cv2_resize(img, (dw, dh), interp))


def test_process_keeps_the_image_on_the_device():
    """process(): HWC3 -> resize_image -> Canny -> control tensor with a non-identity resize (the pipeline has no host image op)"""
    from stablediffusioneo_amd.annotator.canny import CannyDetector
    from stablediffusioneo_amd.annotator.util import HWC3, resize_image
    img = rnd(200, 300, 3, 5)
    dimg = resize_image(torch.from_numpy(HWC3(img)).cuda(), 256)
    assert dimg.is_cuda and tuple(dimg.shape) == (256, 384, 3)
    ctrl = CannyDetector().control_hint(dimg, 100, 200)
    assert ctrl.is_cuda and tuple(ctrl.shape) == (3, 256, 384) and float(ctrl.max()) == 1.0
