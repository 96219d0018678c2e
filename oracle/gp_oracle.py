"""CPU ORACLE for the DKL covariance / exact-GP path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference delegates all GP arithmetic to the third-party package gpytorch
(pinned only as ``gpytorch>=1.9.1`` in the reference's setup.py:40 / requirements.txt:15), which is neither
vendored under /root/reference nor installed here, and the reference's own tests at that boundary assert
shapes/types only (test/models/test_dklgpr.py, test/trainers/test_gptrainer.py).  What is restated below are
the closed forms gpytorch documents for the objects the reference instantiates
(atomai/nets/gp.py:41-46, 55-60, 95-106; atomai/trainers/gptrainer.py:285-303):

  RBFKernel(ard)        k = exp(-1/2 sum_d ((x_d - x'_d)/l_d)^2)
  MaternKernel(nu=2.5)  k = (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r),  r = ||(x - x')/l||
  ScaleKernel           s2 * k;   l, s2, noise = softplus(raw) (noise >= 1e-4);  ConstantMean
  ScaleToBounds(-1, 1)  x -> (x - min) * 0.95*(hi-lo)/(max-min) + 0.95*lo   (min/max of the training batch)
  ExactMarginalLogLikelihood = ( -1/2 r^T Khat^-1 r - 1/2 log det Khat - N/2 log 2 pi ) / N
(The KISS-GP interpolation the reference wraps around the base kernel is NOT reproduced: the north_star asks
for the dense tiled builder.)  Pinned instead by float64 known-answer tests (scipy cho_solve), gpytorch's documented
constants, an independent published implementation (scikit-learn's GaussianProcessRegressor with fixed
hyper-parameters: kernels, posterior mean / variance, log marginal likelihood to 1e-9), symmetry / PSD / diagonal
properties and finite differences in tests/test_gp_*.py.
"""
import numpy as np


def softplus(x):
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)


def kernel_matrix(X1, X2, lengthscale, outputscale, kind="rbf"):
    a = np.asarray(X1, np.float64) / lengthscale
    b = np.asarray(X2, np.float64) / lengthscale
    r2 = np.maximum(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1), 0.0)
    if kind == "rbf":
        return outputscale * np.exp(-0.5 * r2)
    r = np.sqrt(r2)
    return outputscale * (1 + np.sqrt(5) * r + 5.0 / 3.0 * r2) * np.exp(-np.sqrt(5) * r)


def scale_to_bounds(x, lo=-1.0, hi=1.0, minmax=None):
    mn, mx = (x.min(), x.max()) if minmax is None else minmax
    return (x - mn) * (0.95 * (hi - lo) / (mx - mn)) + 0.95 * lo


def exact_mll(Z, y, lengthscale, outputscale, noise, mean, kind="rbf"):
    """Per-datum exact marginal log likelihood and alpha = Khat^-1 (y - mean)."""
    from scipy.linalg import cho_factor, cho_solve
    N = len(y)
    K = kernel_matrix(Z, Z, lengthscale, outputscale, kind) + noise * np.eye(N)
    c = cho_factor(K, lower=True)
    r = np.asarray(y, np.float64) - mean
    alpha = cho_solve(c, r)
    logdet = 2.0 * np.log(np.diag(c[0])).sum()
    return (-0.5 * r @ alpha - 0.5 * logdet - 0.5 * N * np.log(2 * np.pi)) / N, alpha


def posterior(Z, y, Zs, lengthscale, outputscale, noise, mean, kind="rbf"):
    """Latent-function posterior mean and variance at Zs."""
    from scipy.linalg import cho_factor, cho_solve
    N = len(y)
    K = kernel_matrix(Z, Z, lengthscale, outputscale, kind) + noise * np.eye(N)
    c = cho_factor(K, lower=True)
    Ks = kernel_matrix(Z, Zs, lengthscale, outputscale, kind)
    mu = mean + Ks.T @ cho_solve(c, np.asarray(y, np.float64) - mean)
    var = outputscale - np.einsum("ij,ij->j", Ks, cho_solve(c, Ks))
    return mu, var
