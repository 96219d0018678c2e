"""DKL covariance builder A/B (dev tool): AMX_KM_NT bit 0 = streaming stores, bit 1 = hardware exp; torch fill_ of the
same 1 GB matrix as the write-bandwidth reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
ref = None
for nt in ("0", "1", "2", "3", "0", "3"):
    os.environ["AMX_KM_NT"] = nt
    ms = timed(lambda: kernel_matrix(Z, Z, ls, 0.6931, 0))
    Kc = kernel_matrix(Z, Z, ls, 0.6931, 0)
    if ref is None: ref = Kc
    print(f"AMX_KM_NT={nt}: {ms:.4f} ms  {Kc.numel()*4/ms/1e9:.2f} TB/s   max|K - K(nt=0)| {float((Kc - ref).abs().max()):.2e}", flush=True)
