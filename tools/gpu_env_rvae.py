"""rVAE training step (config 4) under the tile-size switches of the 128-unit decoder kernels (dev tool):
AMX_RDEC_FWD_MT in {64, 128}, AMX_RDEC_BWD_MT in {64, 32}; interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
import atomai_amd as aoi

B = 512
rs = np.random.RandomState(0)
X = rs.rand(B * 2, 64, 64).astype(np.float32)
m = aoi.models.rVAE((64, 64), latent_dim=2, seed=0)
m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
m.compile_trainer((X, None), None, batch_size=B)
xs = [torch.from_numpy(X[i * B:(i + 1) * B]).cuda() for i in range(2)]


def step(i):
    m.optim.zero_grad()
    elbo = m.forward_compute_elbo(xs[i % 2])
    (-elbo).backward()
    m.optim.step()
    return elbo


variants = [(64, 64), (128, 64), (64, 32), (128, 32)]
res = {v: [] for v in variants}
for rep in range(3):
    for f, b in variants:
        L.set_knob("AMX_RDEC_FWD_MT", f)
        L.set_knob("AMX_RDEC_BWD_MT", b)
        for i in range(3): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): last = step(i)
        torch.cuda.synchronize()
        res[(f, b)].append((time.perf_counter() - t0) / 10 * 1e3)
for k, v in res.items():
    print(f"fwd MT {k[0]:3d}, bwd MT {k[1]:2d}: {min(v):.3f} ms/step = {B / min(v) * 1e3:.0f} patches/s  ({['%.3f' % t for t in v]})  elbo {float(last):.4f}", flush=True)
