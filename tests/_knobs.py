"""Test helper: change a frozen library switch (csrc/knobs.hip) inside a test."""


def set_knob(monkeypatch, name, value):
    """Plan-comparison tests: change a library switch (an AMX_* row of csrc/knobs.hip) for the rest of the test.  The
    library froze its plan at the first launch, so the environment change is followed by amx_knobs_reload; the autouse
    fixture below restores the frozen default plan after the test."""
    from atomai_amd import _lib as L
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    L.reload_knobs()
