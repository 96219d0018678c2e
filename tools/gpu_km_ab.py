import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd.nets.gp import kernel_matrix
rs = np.random.RandomState(0)
Z = torch.from_numpy(rs.uniform(-1, 1, (16384, 2)).astype(np.float32)).cuda()
ls = torch.full((2,), 0.6931, device="cuda")
for nt in ("0", "1", "0", "1"):
    os.environ["AMX_KM_NT"] = nt
    for _ in range(3): K = kernel_matrix(Z, Z, ls, 0.6931, 0)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): K = kernel_matrix(Z, Z, ls, 0.6931, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"NT={nt}: {ms:.4f} ms  {K.numel()*4/ms/1e9:.2f} TB/s", flush=True)
