"""DKL covariance builder A/B (dev tool): library variants (lib/libatomai_amd_<name>.so; "" = product) x AMX_KM_NT
(bit 0 = streaming stores, bit 1 = hardware exp); torch fill_ of the same 1 GB matrix as the write-bandwidth reference.
   python tools/gpu_km_ab.py [lib names ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib
from atomai_amd.nets.gp import kernel_matrix
rs = np.random.RandomState(0)
Z = torch.from_numpy(rs.uniform(-1, 1, (16384, 2)).astype(np.float32)).cuda()
ls = torch.full((2,), 0.6931, device="cuda")


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


K = torch.empty(16384, 16384, device="cuda")
ms = timed(lambda: K.fill_(1.0))
print(f"torch fill_ 1 GB: {ms:.4f} ms  {K.numel()*4/ms/1e9:.2f} TB/s", flush=True)
libs = {"": _lib.load()}
ref = None
names = [""] + sys.argv[1:]
for rep in range(2):
    for name in names:
        if name not in libs:
            libs[name] = _lib._bind(ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f"libatomai_amd_{name}.so")))
        _lib._lib = libs[name]
        for nt in ("3",) if rep else ("0", "3"):
            os.environ["AMX_KM_NT"] = nt
            ms = timed(lambda: kernel_matrix(Z, Z, ls, 0.6931, 0))
            Kc = kernel_matrix(Z, Z, ls, 0.6931, 0)
            if ref is None: ref = Kc
            Zd = Z.double(); 
            print(f"lib {name or 'product':8s} AMX_KM_NT={nt}: {ms:.4f} ms  {Kc.numel()*4/ms/1e9:.2f} TB/s   max|K - first| {float((Kc - ref).abs().max()):.2e}", flush=True)
_lib._lib = libs[""]
os.environ.pop("AMX_KM_NT", None)
Zd, lsd = Z.double(), ls.double()
for name in names:
    _lib._lib = libs[name]
    ms = timed(lambda: kernel_matrix(Zd[:8192], Zd, lsd, 0.6931, 0), n=5)
    print(f"lib {name or 'product':8s} fp64 8192 x 16384: {ms:.4f} ms  {8192*16384*8/ms/1e9:.2f} TB/s", flush=True)
