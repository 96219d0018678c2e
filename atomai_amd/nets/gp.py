"""Deep-kernel GP modules (reference: atomai/nets/gp.py:14-60).

``fcFeatureExtractor`` keeps the reference's module tree (linear1, relu1, linear2, ...).  The GP layer is an
EXACT Gaussian process on the embeddings whose covariance (and its gradient) is evaluated by the tiled HIP
kernels of csrc/kernel_matrix.hip; Cholesky factorisation / triangular solves are library calls
(torch.linalg -> rocSOLVER), as a GP's dense linear algebra is not part of the hot path named by the
north_star.  The reference wraps the same base kernel in gpytorch's KISS-GP grid interpolation; gpytorch is
not vendored, so this layer follows gpytorch's documented closed forms (oracle/gp_oracle.py).
"""
import math
from typing import Type

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L


class fcFeatureExtractor(nn.Sequential):
    """MLP feature extractor: indim -> 1000 -> 500 -> 50 -> embedim with ReLU (gp.py:14-26)."""

    def __init__(self, feat_dim, embedim, **kwargs):
        super().__init__()
        hidden_dim = kwargs.get("hidden_dim")
        hidden_dim = [1000, 500, 50] if hidden_dim is None else list(hidden_dim)
        hidden_dim.append(embedim)
        self.add_module("linear1", nn.Linear(feat_dim, hidden_dim[0]))
        for i, h in enumerate(hidden_dim[1:]):
            self.add_module('relu{}'.format(i + 1), nn.ReLU())
            self.add_module('linear{}'.format(i + 2), nn.Linear(hidden_dim[i], h))

    def forward(self, x):
        from ._linear import run_dense
        return run_dense(self, x)               # Linear + ReLU pairs fused on the MFMA GEMM (fp64: library)


class convFeatureExtractor(nn.Module):
    """Convolutional feature extractor for flattened square patches (BASELINE.json configs[4]): the reference
    accepts any ``feature_extractor(input_dim, embedim)`` (gptrainer.py:279-280) but ships none with
    convolutions; this one is built from the HIP ConvBlock.  (N, p*p) -> (N, embedim)."""

    def __init__(self, feat_dim, embedim, nb_filters: int = 16):
        super().__init__()
        from .blocks import ConvBlock
        p = int(round(math.sqrt(feat_dim)))
        if p * p != feat_dim or p % 4:
            raise ValueError("convFeatureExtractor needs flattened square patches with side divisible by 4")
        self.p = p
        self.c1 = ConvBlock(2, 1, 1, nb_filters, batch_norm=True)
        self.c2 = ConvBlock(2, 1, nb_filters, 2 * nb_filters, batch_norm=True)
        self.fc = nn.Linear(2 * nb_filters * (p // 4) ** 2, embedim)

    def _apply(self, fn, *args, **kwargs):
        """``.to(torch.float64)`` (dklGPTrainer's precision='double') converts the GP-facing Linear only: the HIP
        convolution blocks compute in fp32 whatever precision the GP layer runs at."""
        super()._apply(fn, *args, **kwargs)
        for blk in (self.c1, self.c2):
            for t in list(blk.parameters()) + list(blk.buffers()):
                if t.is_floating_point() and t.dtype != torch.float32:
                    t.data = t.data.float()
        return self

    def forward(self, x):
        dt = x.dtype
        h = x.reshape(-1, 1, self.p, self.p).float()
        h = F.max_pool2d(self.c1(h), 2, 2)
        h = F.max_pool2d(self.c2(h), 2, 2)
        from ._linear import linear
        return linear(h.flatten(1).to(self.fc.weight.dtype), self.fc.weight, self.fc.bias).to(dt)


# Phase clock of the fit step (tools/bench_extra.py::bench_dkl_fit): None in product use; a dict name -> list of
# (start, end) torch.cuda.Event pairs when a benchmark wants the breakdown (events only, no synchronisation here).
PHASES = None


class _phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PHASES is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PHASES is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PHASES.setdefault(self.name, []).append((self.e0, e1))
        return False


def kernel_matrix(X1, X2, lengthscale, outputscale: float, kind: int = 0, noise: float = 0.0) -> torch.Tensor:
    """Dense K(X1, X2) on the device through the tiled HIP builder (no autograd)."""
    X1, X2 = X1.detach().contiguous(), X2.detach().contiguous()
    N, D = X1.shape
    M = X2.shape[0]
    K = torch.empty(N, M, dtype=X1.dtype, device=X1.device)
    inv_ls = (1.0 / lengthscale.detach().reshape(-1)).to(X1.dtype).contiguous()
    L.call("amx_kernel_matrix", L.ptr(X1), L.ptr(X2), L.ptr(inv_ls), float(outputscale), kind, float(noise), N, M,
           D, int(X1.dtype == torch.float64), L.ptr(K), L.stream_ptr(X1))
    return K


def kernel_matvec(X1, X2, lengthscale, outputscale: float, V, kind: int = 0) -> torch.Tensor:
    """K(X1, X2) @ V (V: M x R, R <= 4) without materialising K."""
    X1, X2, V = X1.detach().contiguous(), X2.detach().contiguous(), V.detach().contiguous()
    N, D = X1.shape
    M, R = V.shape
    Y = torch.empty(N, R, dtype=X1.dtype, device=X1.device)
    inv_ls = (1.0 / lengthscale.detach().reshape(-1)).to(X1.dtype).contiguous()
    L.call("amx_kernel_matvec", L.ptr(X1), L.ptr(X2), L.ptr(inv_ls), float(outputscale), kind, N, M, D, R,
           int(X1.dtype == torch.float64), L.ptr(V), L.ptr(Y), L.stream_ptr(X1))
    return Y


class _ExactMLLFn(torch.autograd.Function):
    """(Z, y, lengthscale, outputscale, noise, mean) -> per-datum exact marginal log likelihood.
    Forward: HIP covariance builder + Cholesky (library).  Backward: G = dMLL/dK = (alpha alpha^T - K^-1)/(2N)
    contracted with the kernel derivatives by the HIP backward kernel (K is recomputed tile-wise)."""

    @staticmethod
    def forward(ctx, Z, y, lengthscale, outputscale, noise, mean, kind):
        N, D = Z.shape
        with _phase("k_build"):
            K = kernel_matrix(Z, Z, lengthscale, float(outputscale), kind, float(noise))
        with _phase("potrf"):
            Lc = torch.linalg.cholesky(K)
        del K
        with _phase("solve_logdet"):
            r = (y.detach() - mean.detach()).reshape(N, 1)
            alpha = torch.cholesky_solve(r, Lc)
            logdet = 2.0 * torch.log(torch.diagonal(Lc)).sum()
            mll = (-0.5 * (r * alpha).sum() - 0.5 * logdet - 0.5 * N * math.log(2 * math.pi)) / N
        ctx.save_for_backward(Z.detach(), lengthscale.detach(), Lc, alpha)
        ctx.meta = (float(outputscale), kind, N, D)
        return mll

    @staticmethod
    def backward(ctx, g):
        Z, ls, Lc, alpha = ctx.saved_tensors
        s2, kind, N, D = ctx.meta
        # K^-1 from the factor (LAPACK potri through torch; the result is symmetric: torch mirrors the computed triangle).
        # G = dMLL/dK = (alpha alpha^T - K^-1) / (2N) is NOT materialised: the HIP backward kernel forms it while it reads
        # K^-1 (round 6: the dense alpha alpha^T, the difference, the scaling and a symmetrised copy were four N x N
        # temporaries and ~10 N^2 memory passes per step)
        with _phase("potri"):
            Kinv = torch.cholesky_inverse(Lc)
            if not Kinv.is_contiguous():        # LAPACK hands back column-major storage: K^-1 is symmetric, read it as it lies
                Kinv = Kinv.T if Kinv.T.is_contiguous() else Kinv.contiguous()
        with _phase("k_bwd"):
            dZ = torch.empty_like(Z)
            nblk = (N + 3) // 4
            part = torch.empty(nblk, D + 1, dtype=Z.dtype, device=Z.device)
            inv_ls = (1.0 / ls.reshape(-1)).to(Z.dtype).contiguous()
            av = alpha.reshape(-1).contiguous()
            L.call("amx_kernel_matrix_bwd_mll", L.ptr(Z.contiguous()), L.ptr(inv_ls), s2, kind, N, D,
                   int(Z.dtype == torch.float64), L.ptr(Kinv), L.ptr(av), 0.5 / N, L.ptr(dZ), L.ptr(part),
                   L.stream_ptr(Z))
            tot = part.sum(0)                                   # (D+1)-vector: plumbing-size reduction
            d_inv_ls, d_s2 = tot[:D], tot[D]
            d_ls = (-d_inv_ls * inv_ls * inv_ls).reshape(ls.shape)
            d_noise = ((av * av).sum() - torch.diagonal(Kinv).sum()) * (0.5 / N)      # trace(G)
        d_mean = alpha.sum() / N                             # d mll / d mean =  sum(alpha) / N
        d_y = -alpha.reshape(-1) / N                         # d mll / d y    = -alpha / N
        return g * dZ, g * d_y, g * d_ls, g * d_s2, g * d_noise, g * d_mean, None


class GPRegressionModel(nn.Module):
    """DKL GP regression module: feature extractor -> ScaleToBounds(-1, 1) -> ConstantMean + ScaleKernel(RBF
    with ARD lengthscales) for each of the q outputs sharing the embedding (gp.py:29-60)."""

    def __init__(self, X: torch.Tensor, y: torch.Tensor, feature_extractor: Type[nn.Module], embedim: int,
                 kernel: str = "rbf") -> None:
        super().__init__()
        q = y.shape[0]
        self.train_inputs, self.train_targets = (X,), y
        self.feature_extractor = feature_extractor
        self.kind = {"rbf": 0, "matern": 1}[kernel]
        dt = X.dtype
        self.raw_lengthscale = nn.Parameter(torch.zeros(q, 1, embedim, dtype=dt))
        self.raw_outputscale = nn.Parameter(torch.zeros(q, dtype=dt))
        self.raw_noise = nn.Parameter(torch.zeros(q, 1, dtype=dt))
        self.mean_constant = nn.Parameter(torch.zeros(q, 1, dtype=dt))
        self.register_buffer("min_val", torch.tensor(0.0, dtype=dt))
        self.register_buffer("max_val", torch.tensor(1.0, dtype=dt))
        self._cache = None

    # gpytorch conventions: softplus transforms, noise >= 1e-4
    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale)

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)

    @property
    def noise(self):
        return F.softplus(self.raw_noise) + 1e-4

    def scale_to_bounds(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            self.min_val.data, self.max_val.data = x.min().detach(), x.max().detach()
            mn, mx = x.min(), x.max()
        else:
            mn, mx = self.min_val, self.max_val
            x = x.clamp(mn, mx)
        return (x - mn) * (0.95 * 2.0 / (mx - mn)) + 0.95 * (-1.0)

    def embed(self, x: torch.Tensor) -> torch.Tensor:
        return self.scale_to_bounds(self.feature_extractor(x))

    def mll(self) -> torch.Tensor:
        """Sum over the q outputs of the per-datum exact marginal log likelihood of the training data."""
        with _phase("extractor_fwd"):
            Z = self.embed(self.train_inputs[0])
        self._cache = None
        tot = 0
        for i in range(self.train_targets.shape[0]):
            tot = tot + _ExactMLLFn.apply(Z, self.train_targets[i], self.lengthscale[i], self.outputscale[i],
                                          self.noise[i, 0], self.mean_constant[i, 0], self.kind)
        return tot

    def _state_key(self) -> tuple:
        """Identity of everything the training-set factorisation depends on: parameter / buffer versions (in-place
        optimizer updates bump them), the fused optimizer's generation counter (it writes through raw pointers),
        the training data and the train / eval mode."""
        from ..engine import _weight_generation
        vers = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        X, y = self.train_inputs[0], self.train_targets
        return (vers, _weight_generation[0], X.data_ptr(), X._version, tuple(X.shape), y.data_ptr(), y._version,
                self.training)

    @torch.no_grad()
    def _posterior_factors(self):
        """(Z_train, [(cholesky(K + noise I), alpha = K^-1 (y - mu)) per output]) — computed ONCE per model state
        and reused by every predict batch (dklgpr.py:202-217 calls the posterior per DataLoader batch; without the
        cache each batch redid the O(N^3) factorisation)."""
        key = self._state_key()
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1], self._cache[2]
        Z = self.embed(self.train_inputs[0])
        factors = []
        for i in range(self.train_targets.shape[0]):
            ls, s2, nz, mu = self.lengthscale[i], float(self.outputscale[i]), float(self.noise[i, 0]), self.mean_constant[i, 0]
            K = kernel_matrix(Z, Z, ls, s2, self.kind, nz)
            Lc = torch.linalg.cholesky(K)
            del K
            alpha = torch.cholesky_solve((self.train_targets[i] - mu).reshape(-1, 1), Lc)
            factors.append((Lc, alpha))
        self._cache = (self._state_key(), Z, factors)        # key re-read: embed() in train mode touches min/max
        self.n_factorisations = getattr(self, "n_factorisations", 0) + 1
        return Z, factors

    @torch.no_grad()
    def posterior(self, x_new: torch.Tensor, full_cov: bool = False):
        """Latent posterior mean (q, n) and variance (q, n) [or covariance (q, n, n)] at x_new."""
        Z, factors = self._posterior_factors()
        Zs = self.embed(x_new)
        means, vars_ = [], []
        for i in range(self.train_targets.shape[0]):
            ls, s2, mu = self.lengthscale[i], float(self.outputscale[i]), self.mean_constant[i, 0]
            Lc, alpha = factors[i]
            means.append(mu + kernel_matvec(Zs, Z, ls, s2, alpha, self.kind).reshape(-1))
            Ks = kernel_matrix(Z, Zs, ls, s2, self.kind)
            v = torch.cholesky_solve(Ks, Lc)
            if full_cov:
                vars_.append(kernel_matrix(Zs, Zs, ls, s2, self.kind) - Ks.T @ v)
            else:
                vars_.append((s2 - (Ks * v).sum(0)).clamp_min(0))
        return torch.stack(means), torch.stack(vars_)


class GPModelList(nn.Module):
    """q INDEPENDENT DKL GPs, one per output row of ``y``, each with its own feature extractor (the reference's
    ``gpytorch.models.IndependentModelList`` of ``GPRegressionModel``s: trainers/gptrainer.py:181-243).  ``models[i]``
    is a single-output ``GPRegressionModel``; the training objective is the sum of their marginal log likelihoods
    (``SumMarginalLogLikelihood``)."""

    def __init__(self, models) -> None:
        super().__init__()
        self.models = nn.ModuleList(models)

    @property
    def train_targets(self):
        return [m.train_targets for m in self.models]

    @property
    def train_inputs(self):
        return [m.train_inputs for m in self.models]

    def mll(self) -> torch.Tensor:
        tot = 0
        for m in self.models:
            tot = tot + m.mll()
        return tot

    def posterior(self, x_new: torch.Tensor, full_cov: bool = False):
        """Stacked latent posterior of the q models at the SAME points: mean (q, n), variance (q, n) [or (q, n, n)]."""
        outs = [m.posterior(x_new, full_cov) for m in self.models]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

    def embed(self, x: torch.Tensor) -> torch.Tensor:
        """(q, n, embedim): every model's own embedding of x."""
        return torch.stack([m.embed(x) for m in self.models])
