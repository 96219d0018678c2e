"""TEST INFRASTRUCTURE: builds (if stale) and injects the CPU-emulated kernel library."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_done = None


def use_emulator():
    """Injects the emulator build (once) and returns its ctypes handle (test-only entry points such as
    amx_emu_set_fast_tanh are reached through it)."""
    global _done
    if _done is not None:
        return _done
    subprocess.check_call([os.path.join(HERE, "build_emu.sh")], stdout=subprocess.DEVNULL)
    from atomai_amd import _lib
    _done = ctypes.CDLL(os.path.join(HERE, "libatomai_amd_emu.so"))
    _lib._inject_for_tests(_done)
    return _done
