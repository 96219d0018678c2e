"""Shared bodies of the DKL tests (emulator tier / gpu tier)."""
import math

import numpy as np
import torch


def _torch_kernel(X1, X2, ls, s2, kind):
    a, b = X1 / ls, X2 / ls
    r2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    if kind == "rbf":
        return s2 * torch.exp(-0.5 * r2)
    r = torch.sqrt(r2 + 1e-300)
    return s2 * (1 + math.sqrt(5) * r + 5.0 / 3.0 * r2) * torch.exp(-math.sqrt(5) * r)


def check_kernel_matrix(device, dtype, kind, N, M, D):
    from oracle import gp_oracle as go
    from atomai_amd.nets.gp import kernel_matrix, kernel_matvec
    rs = np.random.RandomState(N + M)
    X1, X2 = rs.uniform(-1, 1, (N, D)), rs.uniform(-1, 1, (M, D))
    ls, s2 = rs.uniform(0.3, 1.5, D), 1.7
    k = {"rbf": 0, "matern": 1}[kind]
    t = lambda a: torch.from_numpy(a).to(dtype).to(device)
    K = kernel_matrix(t(X1), t(X2), t(ls), s2, k).cpu().numpy()
    ref = go.kernel_matrix(X1, X2, ls, s2, kind)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    np.testing.assert_allclose(K, ref, rtol=tol, atol=tol)
    Ksym = kernel_matrix(t(X1), t(X1), t(ls), s2, k, noise=0.25).cpu().numpy()          # symmetric, noise on diag
    np.testing.assert_allclose(Ksym, Ksym.T, rtol=0, atol=tol)
    np.testing.assert_allclose(np.diag(Ksym), s2 + 0.25, rtol=tol)
    assert np.linalg.eigvalsh(Ksym.astype(np.float64)).min() > 0                          # PSD
    V = rs.randn(M, 3)
    Y = kernel_matvec(t(X1), t(X2), t(ls), s2, t(V), k).cpu().numpy()
    np.testing.assert_allclose(Y, ref @ V, rtol=50 * tol, atol=50 * tol)


def check_mll_and_grads(device, kind, N, D):
    from oracle import gp_oracle as go
    from atomai_amd.nets.gp import _ExactMLLFn
    rs = np.random.RandomState(3)
    Z = torch.from_numpy(rs.uniform(-1, 1, (N, D))).to(device).requires_grad_(True)
    y = torch.from_numpy(np.sin(3 * rs.uniform(-1, 1, N))).to(device)
    ls = torch.tensor(rs.uniform(0.4, 1.0, D), device=device).requires_grad_(True)
    s2 = torch.tensor(1.3, dtype=torch.float64, device=device, requires_grad=True)
    nz = torch.tensor(0.05, dtype=torch.float64, device=device, requires_grad=True)
    mu = torch.tensor(0.1, dtype=torch.float64, device=device, requires_grad=True)
    k = {"rbf": 0, "matern": 1}[kind]
    mll = _ExactMLLFn.apply(Z, y, ls, s2, nz, mu, k)
    ref, _ = go.exact_mll(Z.detach().cpu().numpy(), y.cpu().numpy(), ls.detach().cpu().numpy(), 1.3, 0.05, 0.1, kind)
    assert abs(mll.item() - ref) < 1e-10 * max(1, abs(ref))
    mll.backward()
    # reference gradients: torch autograd through the closed form + torch.linalg (fp64)
    Z2, ls2, s22, nz2, mu2 = (v.detach().clone().requires_grad_(True) for v in (Z, ls, s2, nz, mu))
    K = _torch_kernel(Z2, Z2, ls2, s22, kind) + nz2 * torch.eye(N, dtype=torch.float64, device=device)
    Lc = torch.linalg.cholesky(K)
    r = (y - mu2).reshape(-1, 1)
    alpha = torch.cholesky_solve(r, Lc)
    ref_t = (-0.5 * (r * alpha).sum() - torch.log(torch.diagonal(Lc)).sum() - 0.5 * N * math.log(2 * math.pi)) / N
    ref_t.backward()
    for a, b, name in ((Z, Z2, "Z"), (ls, ls2, "ls"), (s2, s22, "s2"), (nz, nz2, "noise"), (mu, mu2, "mean")):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-7, atol=1e-10, err_msg=name)


def check_dklgpr_api():
    """Shapes / types the reference's tests assert (test/models/test_dklgpr.py)."""
    import atomai_amd as aoi
    rs = np.random.RandomState(0)
    X, y = rs.randn(40, 12), rs.randn(40)
    m = aoi.models.dklGPR(12, embedim=2, precision="double")
    m.fit(X, y, training_cycles=3)
    assert len(m.train_loss) == 3 and all(isinstance(v, float) for v in m.train_loss)
    Xn = rs.randn(15, 12)
    mean, var = m.predict(Xn, batch_size=10)
    assert mean.shape == (15,) and var.shape == (15,) and (var >= 0).all()
    assert m.embed(Xn).shape == (15, 2)
    samples = m.sample_from_posterior(Xn, num_samples=7)
    assert samples.shape == (7, 1, 15)
    ts, idx = m.thompson(Xn)
    assert ts.shape == (1, 15) and idx.shape == (1,)
    ym = rs.randn(2, 40)                                    # two outputs sharing the embedding
    m2 = aoi.models.dklGPR(12, embedim=2, precision="single")
    ls0 = None
    m2.fit(X, ym, training_cycles=2)
    mean, var = m2.predict(Xn)
    assert mean.shape == (2, 15)
    # a lengthscale moves after training (test/trainers/test_gptrainer.py:34-43)
    assert float((m2.gp_model.raw_lengthscale.detach() != 0).float().sum()) > 0
