"""In-process A/B of library variants on the rVAE training step (config 4: 64x64 patches, bs 512) (dev tool).
   python tools/gpu_rvae_ab.py [lib names ...]     ("" = product is always first)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
import atomai_amd as aoi
from atomai_amd.trainers.trainer import _EarlyScalar

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
    early = _EarlyScalar(elbo)
    (-elbo).backward()
    m.optim.step()
    return early.item()


libs = {"": L.load()}
for name in sys.argv[1:]:
    libs[name] = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{name}.so")))
# bit-identity of the builds: one forward + backward on the same weights / input / noise, every gradient compared
torch.manual_seed(7)
eps = torch.randn(B, 5, device="cuda")
m.reparameterize = lambda zm, zs: zm + zs * eps
ref = None
for name, lib in libs.items():
    L._lib = lib
    m.optim.zero_grad()
    elbo = m.forward_compute_elbo(xs[0])
    (-elbo).backward()
    g = [p.grad.detach().clone() for net in (m.encoder_net, m.decoder_net) for p in net.parameters()]
    if ref is None:
        ref = (elbo.item(), g)
    else:
        same = elbo.item() == ref[0] and all(torch.equal(a, b) for a, b in zip(g, ref[1]))
        worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(g, ref[1]))
        print(f"lib {name}: gradients bit-identical to the product build: {same} (worst relative difference {worst:.1e})", flush=True)
del m.reparameterize
m.optim.zero_grad()
res = {k: [] for k in libs}
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for name, lib in libs.items():
        L._lib = lib
        for i in range(3): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): last = step(i)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 10 * 1e3)
for name, v in res.items():
    print(f"lib {name or 'product':8s}: {min(v):.3f} ms/step = {B / min(v) * 1e3:.0f} patches/s   ({['%.3f' % t for t in v]})  elbo {last:.4f}", flush=True)
L._lib = libs[""]
