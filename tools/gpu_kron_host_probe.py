"""Dev tool: host-side cost of the Kronecker core of the KISS-GP layer (nets/gp.py:_SkiCoreKron) on the GPU box's CPU: the
two 50 x 50 eigenproblems through torch / numpy, with and without a one-thread BLAS limit (threadpoolctl)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

G = 50
x = 0.06 * np.arange(G, dtype=np.float64)
K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2)


def t(fn, n=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


out = {"cpus": os.cpu_count(), "torch_threads": torch.get_num_threads()}
out["numpy_eigh_ms"] = t(lambda: np.linalg.eigh(K))
out["torch_eigh_ms"] = t(lambda: torch.linalg.eigh(torch.from_numpy(K)))
try:
    from threadpoolctl import ThreadpoolController
    ctl = ThreadpoolController()
    out["threadpools"] = [(d.get("user_api"), d.get("internal_api"), d.get("num_threads")) for d in ctl.info()]

    def lim():
        with ctl.limit(limits=1):
            np.linalg.eigh(K)
    out["numpy_eigh_1thread_ms"] = t(lim)
except Exception as e:
    out["threadpoolctl"] = repr(e)
V = np.linalg.eigh(K)[1]
a0 = np.arange(300) % G
out["numpy_select_ms"] = t(lambda: np.ascontiguousarray(V[:, a0] * np.sqrt(np.abs(V[0, a0]))))
if torch.cuda.is_available():
    def up():
        a = torch.from_numpy(np.ascontiguousarray(V[:, a0])).cuda()
        b = torch.from_numpy(np.ascontiguousarray(V[:, a0])).cuda()
        F = (a[:, None, :] * b[None, :, :]).reshape(G * G, -1).float().contiguous()
        torch.cuda.synchronize()
    out["upload_and_form_F_ms"] = t(up)
    ls = torch.tensor([0.69, 0.69], device="cuda")
    out["tolist_sync_ms"] = t(lambda: ls.double().tolist())
print(json.dumps(out, indent=1))
