"""ctypes binding of the C-ABI shared library (include/atomai_amd.h).

The product library is ``atomai_amd/lib/libatomai_amd.so`` built by ``__graft_entry__.build()``
with ``hipcc --offload-arch=gfx950``.  It is loaded AFTER torch so that it binds to the HIP runtime
torch already mapped (one ``libamdhip64`` per process, SURVEY.md §7 "Runtime-linking trap").
There is no CPU fallback: if the library is missing, or a tensor is not on a GPU, the ops raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libatomai_amd.so")

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_long
_D = C.c_double

HEADER = os.path.join(os.path.dirname(_HERE), "include", "atomai_amd.h")

_CTYPES = {"int": _I, "long": _L, "float": _F, "double": _D}


def parse_header(path: str = None):
    """Parses the C-ABI header into {name: (restype, [argtypes])}; any pointer -> c_void_p."""
    import re
    text = open(path or HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int|long)\s+(amx_\w+)\s*\(([^)]*)\)\s*;", text):
        res, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                argtypes.append(_P)
            else:
                argtypes.append(_CTYPES[a.split()[-2] if len(a.split()) > 1 else a])
        sigs[name] = (_CTYPES[res], argtypes)
    return sigs


SIGNATURES = parse_header()

_lib = None
_is_test_backend = False


class AmxError(RuntimeError):
    pass


def _bind(cdll):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)          # AttributeError -> symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    cdll.amx_last_error.restype = C.c_char_p     # the two entry points that do not return int
    cdll.amx_last_error.argtypes = []
    cdll.amx_knob_name.restype = C.c_char_p
    cdll.amx_knob_name.argtypes = [_I]
    cdll.amx_knob.argtypes = [C.c_char_p]
    return cdll


def load(path: str = None):
    """Loads (once) and returns the product library."""
    global _lib
    if _lib is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise AmxError(
                f"{path} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        _lib = _bind(C.CDLL(path))
    return _lib


def _inject_for_tests(cdll) -> None:
    """TEST HOOK: tests/emu injects the CPU-emulated build of the same kernel sources so that the
    host logic can be exercised without a GPU.  Never called by the package itself."""
    global _lib, _is_test_backend
    _lib = _bind(cdll)
    _is_test_backend = True


def is_test_backend() -> bool:
    return _is_test_backend


# ---- launch-plan switches (csrc/knobs.hip): the library resolves its AMX_* environment variables ONCE; the host side
# reads the same resolved table, so a switch has one value per process no matter who asks.
_knob_cache = {}
_knob_generation = [-1]


def knob(name: str) -> int:
    """Resolved value of the library switch ``name`` (an AMX_* environment name of csrc/knobs.hip).  Values are cached
    per GENERATION of the library's table (amx_knobs_generation): a reload through any handle of the library — not only
    ``reload_knobs`` below — invalidates the cache, so host and library cannot disagree about a switch."""
    lib = load()
    gen = lib.amx_knobs_generation()
    if gen != _knob_generation[0]:
        _knob_cache.clear()
        _knob_generation[0] = gen
    v = _knob_cache.get(name)
    if v is None:
        v = lib.amx_knob(name.encode())
        if v == -2 ** 31:
            raise AmxError(f"{name} is not a switch of libatomai_amd (see amx_knob_name / DESIGN.md §4)")
        _knob_cache[name] = v
    return v


def knob_names():
    lib = load()
    return [lib.amx_knob_name(i).decode() for i in range(lib.amx_knob_count())]


def set_knob(name: str, value) -> None:
    """A/B scripts and tests: set (or, with None, remove) the environment variable of a switch and have the library
    re-read its table.  Product code never calls this — the plan is frozen at the first launch."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    reload_knobs()


def reload_knobs() -> None:
    _knob_cache.clear()
    load().amx_knobs_reload()


class _StreamPtr(C.c_void_p):
    """hipStream_t of a launch plus the index of the device it belongs to (`call` selects that device)."""
    dev = None


def stream_ptr(t: torch.Tensor):
    if t.is_cuda:
        sp = _StreamPtr(torch.cuda.current_stream(t.device).cuda_stream)
        sp.dev = t.device.index
        return sp
    if not _is_test_backend:
        raise AmxError("atomai_amd ops need tensors on an MI355X (cuda) device; got a CPU tensor "
                       "and there is no CPU fallback")
    return C.c_void_p(0)


class _Ptr(C.c_void_p):
    """Device pointer that keeps its tensor referenced: an argument such as ``L.ptr(x.contiguous())`` must stay alive
    until the launch that reads it has been enqueued (after that the caching allocator's stream order protects it)."""
    keep = None


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "atomai_amd kernels need contiguous tensors"
    assert t.dtype in (torch.float32, torch.int64, torch.uint8, torch.int32, torch.float64)
    p = _Ptr(t.data_ptr())
    p.keep = t
    return p


def last_error() -> str:
    """Message of the last failed call on this thread (amx_last_error of the C ABI)."""
    msg = load().amx_last_error()
    return msg.decode() if msg else ""


def call(name: str, *args):
    fn = getattr(load(), name)
    # A kernel must be launched with ITS device current: the stream in the last argument belongs to the device of
    # the tensors (stream_ptr), which need not be torch's current device (SegPredictor(device='cuda:1'), DKL replicas
    # on ranks > 0).  The C ABI itself never calls hipSetDevice.
    dev = getattr(args[-1], "dev", None) if args else None
    if dev is not None and dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = fn(*args)
    else:
        rc = fn(*args)
    if rc != 0:
        raise AmxError(f"{name} failed with code {rc} "
                       f"({'bad argument #%d' % -rc if rc < 0 else 'hipError_t'}): {last_error()}")
