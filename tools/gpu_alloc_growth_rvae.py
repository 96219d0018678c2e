import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
rs = np.random.RandomState(0)
B = 512
X = rs.rand(B * 2, 64, 64).astype(np.float32)
m = aoi.models.rVAE((64, 64), latent_dim=2, seed=0)
m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
m.compile_trainer((X, None), None, batch_size=B)
xs = [torch.from_numpy(X[i * B:(i + 1) * B]).cuda() for i in range(2)]
def step(i):
    m.optim.zero_grad(); elbo = m.forward_compute_elbo(xs[i % 2]); (-elbo).backward(); m.optim.step(); return elbo.item()
for blk in range(5):
    t0 = time.perf_counter()
    for i in range(10): step(i)
    torch.cuda.synchronize(); s = torch.cuda.memory_stats()
    print(f"steps {10*(blk+1)}: {1e2*(time.perf_counter()-t0):.2f} ms/step reserved {s['reserved_bytes.all.current']/1e9:.2f} GB allocated {s['allocated_bytes.all.current']/1e9:.3f} peak {s['allocated_bytes.all.peak']/1e9:.2f} hipMalloc {s['num_device_alloc']}", flush=True)
