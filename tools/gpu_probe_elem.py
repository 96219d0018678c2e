"""Stand-alone timing of HBM-bound elementwise kernels at the U-Net bs-32 shapes (GB/s of algorithmic bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N = 32
for (h, C) in ((256, 16), (128, 32), (64, 64)):
    du = torch.randn(N, 2 * h, 2 * h, C, device=dev); dv = torch.empty(N, h, h, C, device=dev)
    sp = L.stream_ptr(du)
    ms = timeit(lambda: L.call("amx_upsample2x_bwd", L.ptr(du), L.ptr(dv), N, h, h, C, 0, sp))
    by = (du.numel() + dv.numel()) * 4
    print(f"upsample_bwd h={h} C={C}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s", flush=True)
