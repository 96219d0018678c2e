"""TEST INFRASTRUCTURE: builds (if stale) and injects the CPU-emulated kernel library."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_done = False


def use_emulator():
    global _done
    if _done:
        return
    subprocess.check_call([os.path.join(HERE, "build_emu.sh")], stdout=subprocess.DEVNULL)
    from atomai_amd import _lib
    _lib._inject_for_tests(ctypes.CDLL(os.path.join(HERE, "libatomai_amd_emu.so")))
    _done = True
