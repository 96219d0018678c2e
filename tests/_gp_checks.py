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


# ---------------------------------------------------------------------------------------------------------------
# Known-answer vectors restated from gpytorch's OWN unit tests (the only pin available: gpytorch is neither vendored
# nor installed, so these constants are recalled from its repository, not fetched — the DKL rows stay "parity
# unpinned" in DESIGN.md):
#   test/kernels/test_rbf_kernel.py::TestRBFKernel::test_computes_radial_basis_function
#       a = [4, 2, 8]^T, b = [0, 2]^T, lengthscale 2  ->  exp(-0.5 * [[16, 4], [4, 0], [64, 36]] / 2^2)
#   test/kernels/test_rbf_kernel.py::TestRBFKernel::test_ard
#       a = [[1, 2], [2, 4]], b = [[1, 3], [0, 4]], lengthscales [1, 2]
#   test/kernels/test_matern_kernel.py::TestMaternKernel::test_forward_nu_5_over_2
#       same a, b, lengthscale 2: dist = sqrt(5)/2 * [[4, 2], [2, 0], [8, 6]];  (dist^2/3 + dist + 1) * exp(-dist)
#   gpytorch/utils/grid.py::ScaleToBounds.forward
#       (x - min) * (0.95 * (upper - lower) / (max - min)) + 0.95 * lower, min/max frozen and clamped to in eval mode
GPYTORCH_KAT = {
    "rbf": dict(a=[[4.0], [2.0], [8.0]], b=[[0.0], [2.0]], ls=[2.0],
                K=np.exp(-0.5 * np.array([[16.0, 4.0], [4.0, 0.0], [64.0, 36.0]]) / 4.0)),
    "rbf_ard": dict(a=[[1.0, 2.0], [2.0, 4.0]], b=[[1.0, 3.0], [0.0, 4.0]], ls=[1.0, 2.0],
                    K=np.exp(-0.5 * np.array([[0.0 + 0.25, 1.0 + 1.0], [1.0 + 0.25, 4.0 + 0.0]]))),
    "matern": dict(a=[[4.0], [2.0], [8.0]], b=[[0.0], [2.0]], ls=[2.0], K=None),
}
_d = np.array([[4.0, 2.0], [2.0, 0.0], [8.0, 6.0]]) * (math.sqrt(5) / 2.0)
GPYTORCH_KAT["matern"]["K"] = (_d ** 2 / 3 + _d + 1) * np.exp(-_d)


def check_gpytorch_known_answers_oracle():
    from oracle import gp_oracle as go
    for name, c in GPYTORCH_KAT.items():
        kind = "matern" if name == "matern" else "rbf"
        K = go.kernel_matrix(np.array(c["a"]), np.array(c["b"]), np.array(c["ls"]), 1.0, kind)
        np.testing.assert_allclose(K, c["K"], rtol=1e-14, atol=0, err_msg=name)
    x = np.array([[3.0, -1.0], [0.5, 7.0]])
    np.testing.assert_allclose(go.scale_to_bounds(x), (x + 1.0) * (0.95 * 2 / 8.0) - 0.95, rtol=1e-15)


def check_gpytorch_known_answers_kernel(device):
    from atomai_amd.nets.gp import kernel_matrix
    for name, c in GPYTORCH_KAT.items():
        k = 1 if name == "matern" else 0
        for dt, tol in ((torch.float64, 1e-14), (torch.float32, 1e-6)):
            t = lambda a: torch.tensor(a, dtype=dt, device=device)
            K = kernel_matrix(t(c["a"]), t(c["b"]), t(c["ls"]), 1.0, k).cpu().numpy()
            np.testing.assert_allclose(K, c["K"], rtol=tol, atol=tol * 1e-3, err_msg=f"{name} {dt}")


def check_scale_to_bounds_module(device):
    """GPRegressionModel.scale_to_bounds == gpytorch's ScaleToBounds(-1, 1): train mode records min/max, eval mode
    clamps to them."""
    from atomai_amd.nets.gp import GPRegressionModel
    X = torch.zeros(4, 3, dtype=torch.float64, device=device)
    m = GPRegressionModel(X, torch.zeros(1, 4, dtype=torch.float64, device=device), torch.nn.Identity(), 3).to(device)
    x = torch.tensor([[3.0, -1.0], [0.5, 7.0]], dtype=torch.float64, device=device)
    m.train()
    np.testing.assert_allclose(m.scale_to_bounds(x).cpu().numpy(), ((x.cpu() + 1) * (1.9 / 8) - 0.95).numpy(), rtol=1e-15)
    assert float(m.min_val) == -1.0 and float(m.max_val) == 7.0
    m.eval()
    y = m.scale_to_bounds(torch.tensor([[-5.0, 9.0, 3.0]], dtype=torch.float64, device=device)).cpu().numpy()
    np.testing.assert_allclose(y, [[-0.95, 0.95, 4 * 1.9 / 8 - 0.95]], rtol=1e-15)


def check_posterior_cache(device):
    """predict() in batches factorises the training covariance ONCE (dklgpr.py:202-217 calls the posterior per batch),
    gives the same numbers as one big batch and as the float64 oracle, and the cache dies with the model state."""
    import atomai_amd as aoi
    from oracle import gp_oracle as go
    rs = np.random.RandomState(1)
    X, y = rs.randn(60, 6), np.sin(rs.randn(60))
    m = aoi.models.dklGPR(6, embedim=2, precision="double", device=device, gp="exact")
    m.fit(X, y, training_cycles=2)
    Xn = rs.randn(25, 6)
    gm = m.gp_model
    gm.n_factorisations = 0
    mean_b, var_b = m.predict(Xn, batch_size=4)             # 7 batches
    assert gm.n_factorisations == 1
    mean_1, var_1 = m.predict(Xn)
    assert gm.n_factorisations == 1                         # still the cached factor
    np.testing.assert_allclose(mean_b, mean_1, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var_b, var_1, rtol=1e-9, atol=1e-12)
    Z, Zs = m.embed(X), m.embed(Xn)
    mu, var = go.posterior(Z, y, Zs, gm.lengthscale[0].detach().cpu().numpy().reshape(-1), float(gm.outputscale[0]),
                           float(gm.noise[0, 0]), float(gm.mean_constant[0, 0]), "rbf")
    np.testing.assert_allclose(mean_1, mu, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var_1, var, rtol=1e-7, atol=1e-10)
    with torch.no_grad():                                   # any parameter update invalidates
        gm.raw_noise.add_(0.3)
    m.predict(Xn, batch_size=10)
    assert gm.n_factorisations == 2
    m.fit(X, y, training_cycles=1)                          # so does training
    m.predict(Xn)
    assert gm.n_factorisations == 3


def _stock_conv_extractor(fe):
    """Stock-torch float64 graph with the weights of a convFeatureExtractor: the reference's ConvBlock order
    conv -> LeakyReLU(0.01) -> BatchNorm (atomai/nets/blocks.py:61-76), max-pool, Linear."""
    import torch.nn as nn
    nf = fe.c1.block[0].weight.shape[0]

    class Ref(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Sequential(nn.Conv2d(1, nf, 3, padding=1), nn.LeakyReLU(0.01), nn.BatchNorm2d(nf))
            self.c2 = nn.Sequential(nn.Conv2d(nf, 2 * nf, 3, padding=1), nn.LeakyReLU(0.01), nn.BatchNorm2d(2 * nf))
            self.fc = nn.Linear(fe.fc.in_features, fe.fc.out_features)

        def forward(self, x):
            p = int(round(math.sqrt(x.shape[1])))
            h = x.reshape(-1, 1, p, p)
            h = torch.nn.functional.max_pool2d(self.c1(h), 2, 2)
            h = torch.nn.functional.max_pool2d(self.c2(h), 2, 2)
            return self.fc(h.flatten(1))
    ref = Ref().double()
    sd = {k.replace(".block.", "."): v.detach().cpu().double() for k, v in fe.state_dict().items()}
    ref.load_state_dict(sd)
    return ref


def check_conv_feature_extractor(device, N=24, p=8, nf=4):
    """convFeatureExtractor (BASELINE.json configs[4]'s "conv feature extractor") forward + every gradient, train and
    eval mode, against the stock-torch float64 graph of the same weights."""
    from atomai_amd.nets.gp import convFeatureExtractor
    torch.manual_seed(0)
    fe = convFeatureExtractor(p * p, 2, nb_filters=nf)
    ref = _stock_conv_extractor(fe)
    fe = fe.to(device)
    rs = np.random.RandomState(5)
    x = rs.randn(N, p * p).astype(np.float32)
    w = rs.randn(N, 2)
    for mode in ("train", "eval"):
        getattr(fe, mode)()
        getattr(ref, mode)()
        fe.zero_grad()
        ref.zero_grad()
        xt = torch.from_numpy(x).to(device).requires_grad_(True)
        xr = torch.from_numpy(x).double().requires_grad_(True)
        out, outr = fe(xt), ref(xr)
        np.testing.assert_allclose(out.detach().cpu().numpy(), outr.detach().numpy(), rtol=1e-4, atol=1e-5)
        if mode == "eval":
            continue                                         # backward through eval-mode BatchNorm is not on the path
        (out * torch.from_numpy(w).float().to(device)).sum().backward()
        (outr * torch.from_numpy(w)).sum().backward()
        gref = {k.replace(".", ".block.", 1) if k[:2] in ("c1", "c2") else k: v for k, v in
                ((k, p_.grad) for k, p_ in ref.named_parameters())}
        gmax = max(float(g.abs().max()) for g in gref.values())
        for k, p_ in fe.named_parameters():
            err = float((p_.grad.cpu().double() - gref[k]).abs().max()) / gmax
            assert err < 1e-4, (k, err)
        ex = float((xt.grad.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max())
        assert ex < 1e-4, ex
    # running statistics followed the reference's BatchNorm update
    for k in ("c1.block.2.running_mean", "c2.block.2.running_var"):
        kr = k.replace(".block.", ".")
        np.testing.assert_allclose(fe.state_dict()[k].cpu().numpy(), ref.state_dict()[kr].numpy(), rtol=1e-4, atol=1e-6)


def check_dklgpr_conv_extractor(device, N=256, p=8, cycles=2, precision="single", gp="kissgp"):
    """dklGPR(feature_extractor=convFeatureExtractor) fit + predict (config 5's shape family): loss finite and
    decreasing over the first cycles, predictions finite, variance within [0, s2], one factorisation per predict."""
    import atomai_amd as aoi
    from atomai_amd.nets.gp import convFeatureExtractor
    rs = np.random.RandomState(0)
    X = rs.randn(N, p * p).astype(np.float32)
    y = np.tanh(X[:, : p].sum(1)).astype(np.float32)
    m = aoi.models.dklGPR(p * p, embedim=2, precision=precision, device=device, gp=gp)
    m.fit(X, y, training_cycles=cycles, feature_extractor=convFeatureExtractor)
    assert len(m.train_loss) == cycles and all(np.isfinite(m.train_loss))
    m.gp_model.n_factorisations = 0
    mean, var = m.predict(X[: min(N, 2048)], batch_size=max(64, N // 8))
    assert m.gp_model.n_factorisations == 1
    assert np.isfinite(mean).all() and np.isfinite(var).all()
    s2 = float(m.gp_model.outputscale[0])
    assert (var >= 0).all() and (var <= s2 * (1 + 1e-4)).all()
    return m


def check_extractor_golden(device):
    """fcFeatureExtractor vs tests/golden/gp_extractor.npz (generated by the real reference, oracle/make_golden.py gp):
    state-dict keys, RNG-order initialisation as dklGPTrainer draws it (seed 42, in the trainer's precision), forward
    and every gradient in fp32 (MFMA GEMM path) and fp64 (library path), each within the golden's own fp32 floor."""
    import os
    from atomai_amd.trainers import dklGPTrainer
    from atomai_amd.nets.gp import fcFeatureExtractor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gp_extractor.npz"))

    def mom(v):
        v = v.detach().double().flatten().cpu()
        return np.array([v.numel(), v.sum().item(), (v * v).sum().item(), v[0].item(), v[-1].item()])
    for tag in ("dbl", "sgl", "small", "small_sgl"):
        meta = g[f"{tag}|meta"]
        feat, emb, dbl, full, hid = int(meta[0]), int(meta[1]), bool(meta[2]), bool(meta[3]), [int(v) for v in meta[4:]]
        tr = dklGPTrainer(feat, emb, precision="double" if dbl else "single", device=device)
        fx = (lambda i, e: fcFeatureExtractor(i, e, hidden_dim=list(hid))) if tag.startswith("small") else fcFeatureExtractor
        net = tr._build_extractor(fx, feat, emb)
        assert list(net.state_dict().keys()) == list(g[f"{tag}|keys"])
        for k, v in net.state_dict().items():
            assert v.dtype == (torch.float64 if dbl else torch.float32)
            ref = g[f"{tag}|init|{k}"]
            if full:
                assert np.array_equal(v.cpu().numpy(), ref), (tag, k)          # bit-equal RNG-order init
            else:
                np.testing.assert_allclose(mom(v), ref, rtol=1e-12, atol=0, err_msg=f"{tag} {k}")
        x, w = g[f"{tag}|x"], g[f"{tag}|w"]
        for dt, dtag in ((torch.float32, "f32"), (torch.float64, "f64")):
            import copy
            n2 = copy.deepcopy(net).to(dt)
            y = n2(torch.from_numpy(x).to(dt).to(device))
            (y * torch.from_numpy(w).to(dt).to(device)).sum().backward()
            y64 = g[f"{tag}|y|f64"]
            floor = np.abs(g[f"{tag}|y|f32"] - y64).max()
            tol = 1e-10 if dt == torch.float64 else max(4 * floor, 1e-6 * np.abs(y64).max())
            assert np.abs(y.detach().cpu().numpy() - y64).max() <= tol, (tag, dtag)
            for k, p in n2.named_parameters():
                r64, r32 = g[f"{tag}|grad|{k}|f64"], g[f"{tag}|grad|{k}|f32"]
                got = p.grad.cpu().numpy() if full else mom(p.grad)
                fl = np.abs(r32 - r64).max()
                tol = 1e-9 * max(1.0, np.abs(r64).max()) if dt == torch.float64 else max(4 * fl, 1e-5 * np.abs(r64).max())
                assert np.abs(got - r64).max() <= tol, (tag, dtag, k, np.abs(got - r64).max(), tol)


# ---------------------------------------------------------------------------------------------------------------
# KISS-GP (the reference's GridInterpolationKernel model, csrc/ski.hip + nets/gp.py:_SkiMLLFn)
def _dense_ski_weights(Z, grid):
    """Dense torch (autograd) restatement of the interpolation weights on `grid` (oracle/gp_oracle.py:_keys_cubic)."""
    g0, invd, U = grid.tensors(torch.float64, Z.device)
    N, D = Z.shape

    def keys(t):
        u = t.abs()
        return torch.where(u <= 1, (1.5 * u - 2.5) * u * u + 1,
                           torch.where(u <= 2, ((-0.5 * u + 2.5) * u - 4) * u + 2, torch.zeros_like(u)))
    W = torch.ones(N, 1, dtype=torch.float64, device=Z.device)
    for d in range(D):
        t = (Z[:, d:d + 1] - g0[d]) * invd[d] - torch.arange(grid.G, dtype=torch.float64, device=Z.device)[None, :]
        W = (W[:, :, None] * keys(t)[:, None, :]).reshape(N, -1)
    return W, U


def check_ski_mll_and_grads(device, kind, N, D, G, q=2):
    """_SkiMLLFn (m x m grid algebra + the HIP gather kernels) == the dense N x N evaluation of the SAME model: value vs
    the numpy oracle, every gradient vs torch autograd through W K_UU W^T + noise I and a Cholesky (fp64)."""
    from oracle import gp_oracle as go
    from atomai_amd.nets.gp import SkiGrid, _SkiMLLFn
    rs = np.random.RandomState(10 * D + G)
    Z = torch.from_numpy(rs.uniform(-0.9, 0.9, (N, D))).to(device).requires_grad_(True)
    Y = torch.from_numpy(np.sin(3 * rs.uniform(-1, 1, (q, N)))).to(device)
    ls = torch.tensor(rs.uniform(0.4, 1.0, (q, 1, D)), device=device, requires_grad=True)
    s2 = torch.tensor(rs.uniform(0.7, 1.5, q), device=device, requires_grad=True)
    nz = torch.tensor(rs.uniform(0.05, 0.3, q), device=device, requires_grad=True)
    mu = torch.tensor(rs.uniform(-0.3, 0.3, q), device=device, requires_grad=True)
    grid = SkiGrid(D, G)
    assert grid.update(Z) and not grid.update(Z)
    k = {"rbf": 0, "matern": 1}[kind]
    mll = _SkiMLLFn.apply(Z, Y, ls, s2, nz, mu, k, grid)
    bounds = grid.bounds
    np.testing.assert_allclose(bounds, go.ski_dynamic_bounds(Z.detach().cpu().numpy(), G), rtol=1e-13)
    ref = sum(go.ski_mll(Z.detach().cpu().numpy(), Y[i].cpu().numpy(), ls[i].detach().cpu().numpy(), float(s2[i]),
                         float(nz[i]), float(mu[i]), kind, G, bounds) for i in range(q))
    assert abs(mll.item() - ref) < 1e-10 * max(1, abs(ref)), (mll.item(), ref)
    mll.backward()
    Z2, ls2, s22, nz2, mu2 = (v.detach().clone().requires_grad_(True) for v in (Z, ls, s2, nz, mu))
    W, U = _dense_ski_weights(Z2, grid)
    tot = 0
    for i in range(q):
        K = W @ _torch_kernel(U, U, ls2[i].reshape(-1), s22[i], kind) @ W.T + nz2[i] * torch.eye(N, dtype=torch.float64, device=device)
        Lc = torch.linalg.cholesky(K)
        r = (Y[i] - mu2[i]).reshape(-1, 1)
        al = torch.cholesky_solve(r, Lc)
        tot = tot + (-0.5 * (r * al).sum() - torch.log(torch.diagonal(Lc)).sum() - 0.5 * N * math.log(2 * math.pi)) / N
    tot.backward()
    for a, b, name in ((Z, Z2, "Z"), (ls, ls2, "ls"), (s2, s22, "s2"), (nz, nz2, "noise"), (mu, mu2, "mean")):
        scale = float(b.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-7, atol=1e-9 * scale, err_msg=name)


def check_ski_gram_is_deterministic_and_ragged(device, dtype):
    """amx_ski_gram on clustered points (most cells empty, one cell holding a third of the points): equals the dense
    W^T W / W^T r, is bit-identical between two calls, for D = 1 and 2."""
    from atomai_amd.nets.gp import SkiGrid, ski_gram, ski_weights
    rs = np.random.RandomState(3)
    for D, G, N in ((1, 12, 90), (2, 8, 150)):
        Zn = rs.uniform(-0.9, 0.9, (N, D))
        Zn[: N // 3] = 0.3 + 1e-3 * rs.randn(N // 3, D)
        Z = torch.from_numpy(Zn).to(dtype).to(device)
        R = torch.from_numpy(rs.randn(3, N)).to(dtype).to(device)
        grid = SkiGrid(D, G)
        grid.update(Z)
        base, w, dw = ski_weights(Z, grid)
        A, b = ski_gram(base, w, R, grid)
        A2, b2 = ski_gram(base, w, R, grid)
        assert torch.equal(A, A2) and torch.equal(b, b2)
        W, _ = _dense_ski_weights(Z.double(), grid)
        tol = 1e-12 if dtype == torch.float64 else 2e-5
        np.testing.assert_allclose(A.double().cpu().numpy(), (W.T @ W).cpu().numpy(), rtol=tol, atol=tol)
        np.testing.assert_allclose(b.double().cpu().numpy(), (R.double() @ W).cpu().numpy(), rtol=tol, atol=10 * tol)
        assert float(w.sum(-1).sub(1).abs().max()) < (1e-12 if dtype == torch.float64 else 1e-5)     # weights sum to 1
        if dtype == torch.float64:                              # derivative of the weights: central differences
            h = 1e-6
            bp, wp, _ = ski_weights(Z + h, grid)
            bm, wm, _ = ski_weights(Z - h, grid)
            same = ((bp == base) & (bm == base))[..., None]      # (a point that crosses a node changes its stencil)
            fd = (wp - wm) / (2 * h)
            err = ((dw - fd).abs() * same).max()
            assert float(err) < 1e-6 * float(dw.abs().max()) and float(same.double().mean()) > 0.9


def check_ski_posterior(device, precision="double"):
    """dklGPR (default gp='kissgp') fit + predict vs the numpy oracle of the same model at the trained hyper-parameters:
    mean, variance, full covariance; batched == unbatched; ONE factorisation per model state; the dynamic grid is rebuilt
    when prediction points leave its tight bounds."""
    import atomai_amd as aoi
    from oracle import gp_oracle as go
    rs = np.random.RandomState(1)
    X, y = rs.randn(80, 6), np.sin(rs.randn(2, 80))
    m = aoi.models.dklGPR(6, embedim=2, precision=precision, device=device)
    m.fit(X, y, training_cycles=2, grid_size=12)
    gm = m.gp_model
    assert gm.gp == "kissgp" and gm.grid.G == 12
    Xn = X[:25]                                             # training points: inside the grid's tight bounds by construction
    gm.n_factorisations = 0
    mean_b, var_b = m.predict(Xn, batch_size=4)
    mean_1, var_1 = m.predict(Xn)
    assert gm.n_factorisations == 1
    tol = 1e-9 if precision == "double" else 2e-4
    np.testing.assert_allclose(mean_b, mean_1, rtol=tol, atol=tol)
    np.testing.assert_allclose(var_b, var_1, rtol=10 * tol, atol=tol)
    # off-sample points (they may leave the tight bounds: the dynamic grid then follows, as gpytorch's does); the oracle is
    # evaluated on the grid the call ended with
    Xn = X[:25] + 0.05 * rs.randn(25, 6)
    mean_1, cov = m._compute_posterior(torch.from_numpy(Xn).to(m.dtype), full_cov=True)
    mean_1 = mean_1.cpu().numpy()
    var_1 = m._compute_posterior(torch.from_numpy(Xn).to(m.dtype))[1].cpu().numpy()
    Z, Zs = m.embed(X).astype(np.float64), m.embed(Xn).astype(np.float64)
    otol = 1e-7 if precision == "double" else 5e-3
    for i in range(2):
        mu, C = go.ski_posterior(Z, y[i], Zs, gm.lengthscale[i].detach().cpu().numpy().reshape(-1), float(gm.outputscale[i]),
                                 float(gm.noise[i, 0]), float(gm.mean_constant[i, 0]), "rbf", 12, gm.grid.bounds)
        np.testing.assert_allclose(mean_1[i], mu, rtol=otol, atol=otol)
        np.testing.assert_allclose(var_1[i], np.diag(C), rtol=10 * otol, atol=otol)
        np.testing.assert_allclose(cov[i].cpu().numpy(), C, rtol=0, atol=10 * otol)
    # points far outside the training range: the grid is rebuilt over the union and the factors follow
    v0 = gm.grid.version
    far = torch.from_numpy(10 * rs.randn(5, 6)).to(m.dtype)
    mf, vf = m._compute_posterior(far)
    assert torch.isfinite(mf).all() and (vf >= 0).all()
    lo, hi = zip(*gm.grid.tight_bounds())
    zf = m.embed(far.cpu().numpy())
    assert (zf >= np.array(lo) - 1e-6).all() and (zf <= np.array(hi) + 1e-6).all() and gm.grid.version >= v0
    samples = m.sample_from_posterior(Xn, num_samples=5)
    assert samples.shape == (5, 2, 25) and np.isfinite(samples).all()


def check_ski_fallbacks():
    """embedim 3 (50^3 grid nodes) and gp='exact' run the dense exact GP, with a warning in the first case."""
    import warnings
    import atomai_amd as aoi
    rs = np.random.RandomState(0)
    X, y = rs.randn(30, 5), rs.randn(30)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        m = aoi.models.dklGPR(5, embedim=3, precision="double")
        m.fit(X, y, training_cycles=1)
    assert m.gp_model.gp == "exact" and any("exact dense GP" in str(w.message) for w in wlist)
    m = aoi.models.dklGPR(5, embedim=2, precision="double", gp="exact")
    m.fit(X, y, training_cycles=1)
    assert m.gp_model.gp == "exact" and m.gp_model.grid is None
    m = aoi.models.dklGPR(5, embedim=1, precision="double")
    m.fit(X, y, training_cycles=2)
    assert m.gp_model.gp == "kissgp" and m.gp_model.grid.m == 50
    mean, var = m.predict(X[:7])
    assert mean.shape == (7,) and (var >= 0).all()


def check_ski_kron_core_equals_lu_core(device, N=300, G=20):
    """The RBF core (Kronecker eigen-decomposition of K_UU, r x r Cholesky) against the general LU core on the same inputs:
    marginal log likelihood, every gradient and the posterior factors; the kept rank is far below the grid size."""
    import atomai_amd.nets.gp as gp
    rs = np.random.RandomState(4)
    Zn = rs.uniform(-0.9, 0.9, (N, 2))
    Y = torch.from_numpy(np.sin(3 * Zn[:, :1].T) + 0.1 * rs.randn(1, N)).to(device)
    grid = gp.SkiGrid(2, G)
    res = []
    for kron in (False, True):
        gp.SKI_KRON[0] = kron
        try:
            Z = torch.from_numpy(Zn).to(device).requires_grad_(True)
            ls = torch.tensor([[[0.7, 0.5]]], dtype=torch.float64, device=device, requires_grad=True)
            s2 = torch.tensor([1.2], dtype=torch.float64, device=device, requires_grad=True)
            nz = torch.tensor([0.07], dtype=torch.float64, device=device, requires_grad=True)
            mu = torch.tensor([0.1], dtype=torch.float64, device=device, requires_grad=True)
            grid.update(Z)
            mll = gp._SkiMLLFn.apply(Z, Y, ls, s2, nz, mu, 0, grid)
            mll.backward()
            res.append([mll.detach()] + [t.grad.clone() for t in (Z, ls, s2, nz, mu)])
        finally:
            gp.SKI_KRON[0] = True
    for a, b in zip(*res):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-7 * float(b.abs().max()))
    base, w, _ = gp.ski_weights(Z.detach(), grid)
    A, b = gp.ski_gram(base, w, Y - 0.1, grid)
    _, _, U = grid.tensors(torch.float64, Z.device)
    ck = gp._SkiCoreKron(U, ls.detach()[0], 1.2, 0, A, b[0], 0.07, grid)
    cl = gp._SkiCoreLU(U, ls.detach()[0], 1.2, 0, A, b[0], 0.07, grid)
    assert ck.r <= grid.m
    for f in ("Q", "PtA", "trP"):
        x, y = getattr(ck, f)(), getattr(cl, f)()
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-7 * float(y.abs().max()), err_msg=f)
    np.testing.assert_allclose(ck.x.cpu().numpy(), cl.x.cpu().numpy(), rtol=1e-6, atol=1e-8 * float(cl.x.abs().max()))
    return ck.r
