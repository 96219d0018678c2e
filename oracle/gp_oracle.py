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
(The KISS-GP interpolation the reference wraps around the base kernel is NOT reproduced by the PRODUCT: the north_star
asks for the dense tiled builder.  Its size is QUANTIFIED at the end of this file — `ski_kernel_matrix`, a numpy
restatement of structured kernel interpolation with gpytorch's published grid conventions — and asserted in
tests/test_gp_emulated.py: at the default grid of 50 points per dimension the interpolated covariance differs from the
exact one by < 1e-3 of the output scale for lengthscales >= 0.3.)  Pinned instead by float64 known-answer tests (scipy cho_solve), gpytorch's documented
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


# ---------------------------------------------------------------------------------------------------------------
# KISS-GP / SKI restatement ("the exact-GP vs KISS-GP difference is documented but not quantified").
# The reference's covariance module is gpytorch.kernels.GridInterpolationKernel(base_kernel, num_dims=embedim,
# grid_size=50) (atomai/nets/gp.py:41-46): structured kernel interpolation, Wilson & Nickisch, "Kernel Interpolation for
# Scalable Structured Gaussian Processes (KISS-GP)", ICML 2015:   K_SKI(X1, X2) = W1 K_UU W2^T,   K_UU the base kernel on
# a regular grid U, W the sparse matrix of local CUBIC-CONVOLUTION interpolation weights (Keys 1981, a = -0.5; 4 grid
# points per dimension, 4^d per data point).  gpytorch's conventions as published in its source (gpytorch/utils/grid.py,
# gpytorch/utils/interpolation.py; the package itself is absent here — this block is as UNPINNED as the rest of the file):
# default grid bounds (-1, 1) per dimension — the range ScaleToBounds(-1, 1) maps the embeddings into —, the grid
# EXTENDED by one spacing on either side:  spacing = (hi - lo) / (grid_size - 2),  points = linspace(lo - spacing,
# hi + spacing, grid_size).  Dense numpy evaluation (no Toeplitz / Kronecker shortcuts: they change cost, not values).
def _keys_cubic(s):
    s = np.abs(s)
    return np.where(s <= 1.0, (1.5 * s - 2.5) * s * s + 1.0,
                    np.where(s < 2.0, ((-0.5 * s + 2.5) * s - 4.0) * s + 2.0, 0.0))


def ski_grid(grid_size=50, bounds=(-1.0, 1.0)):
    spacing = (bounds[1] - bounds[0]) / (grid_size - 2)
    return np.linspace(bounds[0] - spacing, bounds[1] + spacing, grid_size)


def ski_interp_weights(x, grid):
    """Dense (n, grid_size) matrix of 1-D cubic-convolution weights of the points x on the regular grid."""
    h = grid[1] - grid[0]
    return _keys_cubic((np.asarray(x, np.float64)[:, None] - grid[None, :]) / h)


def ski_dynamic_bounds(X, grid_size=50):
    """Grid bounds per dimension as GridInterpolationKernel.forward sets them when none were passed (the reference passes
    none: atomai/nets/gp.py:45-46) and the grid has not been built yet or an input left the tight bounds:
    spacing = (max - min) / (grid_size - 4.02);  bounds = (min - 2.01 spacing, max + 2.01 spacing)."""
    X = np.asarray(X, np.float64)
    out = []
    for mn, mx in zip(X.min(0), X.max(0)):
        sp = (mx - mn) / (grid_size - 4.02)
        out.append((mn - 2.01 * sp, mx + 2.01 * sp))
    return out


def _per_dim_bounds(bounds, D):
    b = np.asarray(bounds, np.float64)
    return [tuple(b)] * D if b.ndim == 1 else [tuple(v) for v in b]


def ski_weights_and_grid(X, grid_size=50, bounds=(-1.0, 1.0)):
    """(W [n][grid_size^D], U [grid_size^D][D]): dense interpolation weights on the product grid; node order i0 * G + i1."""
    X = np.asarray(X, np.float64)
    D = X.shape[1]
    gs = [ski_grid(grid_size, b) for b in _per_dim_bounds(bounds, D)]
    W = np.ones((X.shape[0], 1))
    for d in range(D):                                        # Kronecker structure of the product grid, row-wise
        Wd = ski_interp_weights(X[:, d], gs[d])
        W = (W[:, :, None] * Wd[:, None, :]).reshape(X.shape[0], -1)
    U = np.stack(np.meshgrid(*gs, indexing="ij"), -1).reshape(-1, D)
    return W, U


def ski_kernel_matrix(X1, X2, lengthscale, outputscale, kind="rbf", grid_size=50, bounds=(-1.0, 1.0)):
    """K_SKI(X1, X2) = W1 K_UU W2^T on the product grid of `grid_size` points per embedding dimension (`bounds`: one
    (lo, hi) pair for every dimension or one pair per dimension)."""
    X1, X2 = np.asarray(X1, np.float64), np.asarray(X2, np.float64)
    D = X1.shape[1]
    ls = np.broadcast_to(np.asarray(lengthscale, np.float64).reshape(-1), (D,))
    W1, U = ski_weights_and_grid(X1, grid_size, bounds)
    W2, _ = ski_weights_and_grid(X2, grid_size, bounds)
    Kuu = kernel_matrix(U, U, ls, outputscale, kind)
    return W1 @ Kuu @ W2.T


def ski_mll(Z, y, lengthscale, outputscale, noise, mean, kind="rbf", grid_size=50, bounds=(-1.0, 1.0)):
    """Per-datum marginal log likelihood of y ~ N(mean, K_SKI + noise I): gpytorch's ExactMarginalLogLikelihood on the
    reference's model, by a dense N x N Cholesky (the product evaluates the same number through m x m grid algebra)."""
    from scipy.linalg import cho_factor, cho_solve
    N = len(y)
    K = ski_kernel_matrix(Z, Z, lengthscale, outputscale, kind, grid_size, bounds) + noise * np.eye(N)
    c = cho_factor(K, lower=True)
    r = np.asarray(y, np.float64) - mean
    return (-0.5 * r @ cho_solve(c, r) - np.log(np.diag(c[0])).sum() - 0.5 * N * np.log(2 * np.pi)) / N


def ski_posterior(Z, y, Zs, lengthscale, outputscale, noise, mean, kind="rbf", grid_size=50, bounds=(-1.0, 1.0)):
    """Latent posterior mean and FULL covariance at Zs of the KISS-GP model (every covariance through the interpolation,
    the test-test block included, as GridInterpolationKernel evaluates it)."""
    from scipy.linalg import cho_factor, cho_solve
    N = len(y)
    K = ski_kernel_matrix(Z, Z, lengthscale, outputscale, kind, grid_size, bounds) + noise * np.eye(N)
    Ks = ski_kernel_matrix(Z, Zs, lengthscale, outputscale, kind, grid_size, bounds)
    Kss = ski_kernel_matrix(Zs, Zs, lengthscale, outputscale, kind, grid_size, bounds)
    c = cho_factor(K, lower=True)
    mu = mean + Ks.T @ cho_solve(c, np.asarray(y, np.float64) - mean)
    return mu, Kss - Ks.T @ cho_solve(c, Ks)


def ski_vs_exact(N=400, D=2, lengthscale=0.6931, outputscale=0.6931, kind="rbf", grid_size=50, seed=0):
    """max |K_SKI - K_exact| / outputscale and the relative change of the posterior mean on a smooth target, for N points
    uniform in the ScaleToBounds range [-0.95, 0.95]^D: what the dense GP of this build differs by from the reference's
    covariance module at equal hyper-parameters."""
    rs = np.random.RandomState(seed)
    Z = rs.uniform(-0.95, 0.95, (N, D))
    Zs = rs.uniform(-0.95, 0.95, (64, D))
    Ke = kernel_matrix(Z, Z, lengthscale, outputscale, kind)
    Ks = ski_kernel_matrix(Z, Z, lengthscale, outputscale, kind, grid_size)
    y = np.sin(3 * Z[:, 0]) + 0.1 * rs.randn(N)
    noise = 0.01
    from scipy.linalg import cho_factor, cho_solve
    mu = []
    for K, Kx in ((Ke, kernel_matrix(Zs, Z, lengthscale, outputscale, kind)),
                  (Ks, ski_kernel_matrix(Zs, Z, lengthscale, outputscale, kind, grid_size))):
        c = cho_factor(K + noise * np.eye(N), lower=True)
        mu.append(Kx @ cho_solve(c, y))
    return {"kernel_max_abs_over_s2": float(np.abs(Ks - Ke).max() / outputscale),
            "posterior_mean_max_abs": float(np.abs(mu[0] - mu[1]).max()),
            "posterior_mean_scale": float(np.abs(mu[0]).max())}
