"""The C-ABI library loads and exports every symbol include/atomai_amd.h declares (no compute)."""
import ctypes
import os

import pytest

from atomai_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_parses():
    sigs = _lib.parse_header()
    assert len(sigs) >= 30 and "amx_conv2d_fwd" in sigs and "amx_adam_flat" in sigs


def test_product_library_exports_header():
    import __graft_entry__ as g
    g.build()                                    # incremental: a no-op when the library is up to date
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _lib.parse_header() if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU: CPU tensors are rejected."""
    import torch
    if torch.cuda.is_available() or _lib.is_test_backend():
        pytest.skip("needs a GPU-less process without the test emulator injected")
    with pytest.raises(_lib.AmxError):
        _lib.stream_ptr(torch.zeros(4))
