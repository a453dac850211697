"""`annotator/canny/__init__.py:4-6` on the HIP path: `CannyDetector()(img, low_threshold, high_threshold)` returns what
`cv2.Canny(img, low_threshold, high_threshold)` returns (HxW uint8, 0 / 255), computed by csrc/canny.hip through
`sdeo_canny_u8`.  There is no CPU path: without the library or a HIP device the call raises."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib
from ..._lib import check, cur_stream, ptr


class CannyDetector:
    def _run(self, img, low_threshold, high_threshold, want_edges: bool, want_control: bool):
        lib = _lib.load()
        lib.sdeo_canny_workspace_bytes.restype = C.c_size_t
        if not torch.cuda.is_available():
            raise _lib.SdeoError("CannyDetector needs a HIP device (there is no CPU fallback)")
        t = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
        if t.dtype != torch.uint8:
            raise _lib.SdeoError(f"CannyDetector: uint8 image expected, got {t.dtype}")
        if t.dim() == 2:
            t = t[:, :, None]
        h, w, c = t.shape
        t = t.to("cuda").contiguous()
        edges = torch.empty((h, w), dtype=torch.uint8, device=t.device) if want_edges else None
        control = torch.empty((3, h, w), dtype=torch.float32, device=t.device) if want_control else None
        nb = int(lib.sdeo_canny_workspace_bytes(C.c_int(h), C.c_int(w)))
        ws = torch.empty(nb, dtype=torch.uint8, device=t.device)
        check(lib.sdeo_canny_u8(ptr(t), C.c_int(h), C.c_int(w), C.c_int(c), C.c_float(low_threshold), C.c_float(high_threshold),
                                ptr(edges), ptr(control), ptr(ws), C.c_size_t(nb), cur_stream()), "canny")
        return edges, control

    def __call__(self, img, low_threshold, high_threshold):
        """numpy in -> numpy out (the reference's contract); torch tensor in -> CUDA tensor out."""
        edges, _ = self._run(img, low_threshold, high_threshold, True, False)
        return edges.cpu().numpy() if isinstance(img, np.ndarray) else edges

    def control_hint(self, img, low_threshold, high_threshold):
        """HWC3(edges) / 255 as a (3, H, W) fp32 CUDA tensor (`canny2image_torch.py:34-38`) without leaving the GPU."""
        _, control = self._run(img, low_threshold, high_threshold, False, True)
        return control
