"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/sdeo.h declares (no compute calls -- there is no GPU here)."""
import ctypes

import pytest

from stablediffusioneo_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    names = _lib.declared_symbols()
    assert len(names) >= 25 and "sdeo_groupnorm_nhwc_f16" in names and "sdeo_unet_forward" in names
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_error_string(lib):
    assert lib.sdeo_version() == 101          # include/sdeo.h SDEO_ABI_VERSION; _lib.load() refuses a library that disagrees with the header
    assert isinstance(lib.sdeo_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    """Shape validation runs on the host before any launch, so it is testable without a device."""
    rc = lib.sdeo_layernorm_f16(None, None, None, None, ctypes.c_int(4), ctypes.c_int(64), ctypes.c_float(1e-5), None)
    assert rc != 0 and b"layernorm" in lib.sdeo_last_error()
    rc = lib.sdeo_attention_f16(ctypes.c_void_p(16), 64, ctypes.c_void_p(16), 64, ctypes.c_void_p(16), 64, ctypes.c_void_p(16),
                                64, 1, 1, 8, 8, 8, 8, 12, ctypes.c_float(1.0), None)
    assert rc != 0 and b"head dim" in lib.sdeo_last_error()


def test_product_path_has_no_oracle_import():
    """The oracle is test infrastructure: nothing under stablediffusioneo_amd/ may import it."""
    import os
    import re
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
