"""`annotator/util.py` of the reference: HWC3 and resize_image (same names, same semantics)."""
from ..canny2image import HWC3, resize_image, target_size  # noqa: F401
