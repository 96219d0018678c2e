"""Dev tool: where the ~1 ms of the KISS-GP `ski_gram` phase goes at N = 16384 (nets/gp.py: ski_weights, the cell sort in
torch, amx_ski_gram) — HIP-event times of the pieces."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd.nets.gp as gp
from atomai_amd import _lib as L


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)          # us


out = {}
for name, Zn in (("uniform", np.random.RandomState(0).uniform(-0.95, 0.95, (16384, 2))),
                 ("clustered", np.clip(0.05 * np.random.RandomState(1).randn(16384, 2), -0.95, 0.95))):
    Z = torch.from_numpy(Zn.astype(np.float32)).cuda()
    grid = gp.SkiGrid(2, 50); grid.update(Z)
    R = torch.randn(1, 16384, device="cuda")
    base, w, dw, cell = gp.ski_weights(Z, grid, want_cell=True)
    r = {"ski_weights": t(lambda: gp.ski_weights(Z, grid, want_cell=True)),
         "sort_int32": t(lambda: torch.sort(cell, stable=True)),
         "cells_total": t(lambda: gp._ski_cells(base, 50, cell)),
         "zeros_mxm": t(lambda: torch.zeros(2500, 2500, device="cuda")),
         "ski_gram_total": t(lambda: gp.ski_gram(base, w, R, grid, cell=cell))}
    order, start = gp._ski_cells(base, 50, cell)
    ws = torch.empty((47 ** 2) * (256 + 16), device="cuda"); A = torch.zeros(2500, 2500, device="cuda"); b = torch.empty(1, 2500, device="cuda")
    r["amx_ski_gram_kernels"] = t(lambda: L.call("amx_ski_gram", L.ptr(w), L.ptr(R), L.ptr(order), L.ptr(start), 16384, 2, 50, 1, 0,
                                                  L.ptr(ws), L.ptr(A), L.ptr(b), L.stream_ptr(w)))
    r["max_points_per_cell"] = int((start[1:] - start[:-1]).max())
    out[name] = r
print(json.dumps(out, indent=1))
