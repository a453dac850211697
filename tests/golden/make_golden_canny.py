"""Regression fixture for the Canny path.  PARITY UNPINNED: cv2 is not importable in the build container and the
reference stores no edge map, so the expected output is the oracle's (oracle/canny_oracle.py), frozen here; the INPUT is one of the
reference's own test pictures (`pictures_croped/bird_0.jpg`, 256x384, the data `compute_score.py:44-47` feeds to `process`),
decoded with PIL (cv2.imread's decoder is not available either; only the pixel array is stored, as data).

    PYTHONPATH=/root/repo python tests/golden/make_golden_canny.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import canny_oracle as O  # noqa: E402

img = np.asarray(Image.open("/root/reference/pictures_croped/bird_0.jpg").convert("RGB"))[:, :, ::-1].copy()   # BGR like cv2.imread
assert img.shape == (256, 384, 3) and img.dtype == np.uint8, img.shape
edges = O.canny(img, 100, 200)                    # thresholds of `compute_score.py:48-55`
out = os.path.join(ROOT, "tests", "golden", "canny.npz")
np.savez_compressed(out, image=img, low=np.int32(100), high=np.int32(200), edges=np.packbits(edges > 0))
print(out, os.path.getsize(out), "bytes; edge pixels:", int((edges > 0).sum()))
