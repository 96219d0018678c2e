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


def test_switch_table_is_resolved_once_and_reloadable(monkeypatch):
    """csrc/knobs.hip: every AMX_* switch of the library is a row of one table, resolved on first use; a later change of
    the environment is NOT seen (no getenv on the launch path) until amx_knobs_reload() (A/B scripts, tests)."""
    import __graft_entry__ as g
    g.build()
    lib = _lib._bind(ctypes.CDLL(_lib.LIB_PATH))
    n = lib.amx_knob_count()
    names = [lib.amx_knob_name(i).decode() for i in range(n)]
    assert n >= 10 and len(set(names)) == n and all(x.startswith("AMX_") for x in names)
    assert lib.amx_knob_name(n) is None and lib.amx_knob(b"AMX_NO_SUCH_SWITCH") == -2 ** 31
    monkeypatch.delenv("AMX_CONV_WS", raising=False)
    lib.amx_knobs_reload()
    assert lib.amx_knob(b"AMX_CONV_WS") == 1 and lib.amx_knob(b"AMX_WGRAD_WS_MASK") == 3
    monkeypatch.setenv("AMX_CONV_WS", "0")
    assert lib.amx_knob(b"AMX_CONV_WS") == 1                  # frozen
    lib.amx_knobs_reload()
    assert lib.amx_knob(b"AMX_CONV_WS") == 0
    monkeypatch.delenv("AMX_CONV_WS")
    lib.amx_knobs_reload()
    assert lib.amx_knob(b"AMX_CONV_WS") == 1


def test_no_getenv_on_the_launch_path():
    """The kernel sources read the environment in exactly one place (knobs.hip)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "atomai_amd", "csrc")
    offenders = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        if os.path.basename(f) == "knobs.hip":
            continue
        for i, line in enumerate(open(f), 1):
            if re.search(r"\bgetenv\s*\(", line):
                offenders.append(f"{os.path.basename(f)}:{i}")
    assert not offenders, offenders
