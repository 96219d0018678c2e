"""Dev tool: where the 22 ms of the KISS-GP grid solve go (nets/gp.py:_ski_solve, m = 2500) — torch.linalg pieces timed one
by one with HIP events, default linalg backend and MAGMA if this torch has it."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


m = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
out = {}
for dt in (torch.float32, torch.float64):
    g = torch.Generator(device="cuda").manual_seed(0)
    B = torch.randn(m, m, dtype=dt, device="cuda", generator=g)
    K = B @ B.T / m
    A = torch.randn(m, m, dtype=dt, device="cuda", generator=g); A = A @ A.T / m
    M = K @ A + 0.5 * torch.eye(m, dtype=dt, device="cuda")
    S = K + 0.5 * torch.eye(m, dtype=dt, device="cuda")
    I = torch.eye(m, dtype=dt, device="cuda")
    for lib in ("default", "magma"):
        try:
            torch.backends.cuda.preferred_linalg_library(lib)
        except Exception as e:
            out[f"{dt}|{lib}"] = f"unavailable: {e}"; continue
        r = {}
        try:
            r["gemm"] = t(lambda: K @ A)
            r["lu_factor"] = t(lambda: torch.linalg.lu_factor(M))
            LU, piv = torch.linalg.lu_factor(M)
            r["lu_solve_I"] = t(lambda: torch.linalg.lu_solve(LU, piv, I))
            r["lu_solve_1rhs"] = t(lambda: torch.linalg.lu_solve(LU, piv, I[:, :1]))
            r["inv"] = t(lambda: torch.linalg.inv(M))
            r["solve_I"] = t(lambda: torch.linalg.solve(M, I))
            r["cholesky"] = t(lambda: torch.linalg.cholesky(S))
            Lc = torch.linalg.cholesky(S)
            r["cholesky_inverse"] = t(lambda: torch.cholesky_inverse(Lc))
            r["cholesky_solve_I"] = t(lambda: torch.cholesky_solve(I, Lc))
            r["trsm_lower_I"] = t(lambda: torch.linalg.solve_triangular(Lc, I, upper=False))
            r["eigh"] = t(lambda: torch.linalg.eigh(S), n=2)
        except Exception as e:
            r["error"] = f"{type(e).__name__}: {e}"
        out[f"{dt}|{lib}"] = r
print(json.dumps(out, indent=1))
