"""Deep-kernel GP modules (reference: atomai/nets/gp.py:14-60).

``fcFeatureExtractor`` keeps the reference's module tree (linear1, relu1, linear2, ...).  The GP layer on the embeddings
comes in two forms, both around the tiled HIP covariance builder of csrc/kernel_matrix.hip:

* ``gp="kissgp"`` (default, embedim <= 2): the reference's model — gpytorch's ``GridInterpolationKernel(base_kernel,
  num_dims=embedim, grid_size=50)`` (gp.py:41-46): K = W K_UU W^T with K_UU the base kernel on a regular grid and W cubic
  interpolation weights (csrc/ski.hip).  The marginal log likelihood and the posterior of THAT model are evaluated exactly
  through the matrix-inversion / determinant lemmas on the m = grid_size^embedim grid nodes (O(N + m^3) per step instead of
  the dense O(N^3)); gpytorch reaches the same quantities with CG / stochastic Lanczos estimators (training) and LOVE
  (``fast_pred_var``), i.e. up to ITS solver tolerances.
* ``gp="exact"``: a dense exact GP (Cholesky through torch.linalg -> rocSOLVER); also the fallback for embedim > 2, where
  the dense m x m grid algebra does not fit (gpytorch uses Kronecker / Toeplitz structure there).

gpytorch is not vendored and not installed: both forms follow its published conventions (oracle/gp_oracle.py) and stay
PARITY-UNPINNED.
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

    def _build(self, tape, x):
        """conv -> pool -> conv -> pool as ONE tape (NHWC throughout, pooling on the HIP kernels, the BatchNorm affine applied
        by each consumer): no NCHW round trips and no library pooling between the blocks."""
        node, c1 = self.c1._emit_input(tape, x, pool_next=True)
        c2 = self.c2._emit(tape, [tape.pool(c1)])
        return node, tape.output(tape.pool(c2))

    def forward(self, x):
        dt = x.dtype
        h = x.reshape(-1, 1, self.p, self.p).float()
        from ._function import run_tape
        h = run_tape(self._build, h, list(self.c1.parameters()) + list(self.c2.parameters()), self.training)
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


# ---------------------------------------------------------------------------------------------------------------
# KISS-GP (csrc/ski.hip)
class SkiGrid:
    """The interpolation grid of gpytorch's GridInterpolationKernel when no ``grid_bounds`` are passed (the reference passes
    none, gp.py:45-46): DYNAMIC — (re)built from the per-dimension range of the inputs whenever it has never been built
    or an input falls outside the current 'tight' bounds (GridInterpolationKernel.forward / _tight_grid_bounds):
        spacing = (max - min) / (G - 4.02);   bounds = (min - 2.01 spacing, max + 2.01 spacing)
        tight bounds = (lo + 2.01 (hi - lo) / G, hi - 2.01 (hi - lo) / G)
    and the nodes are gpytorch.utils.grid.create_grid(extend=True): linspace(lo - d, hi + d, G), d = (hi - lo) / (G - 2)."""

    def __init__(self, D: int, G: int):
        self.D, self.G = D, G
        self.bounds = None                   # [(lo, hi)] per dimension
        self.version = 0
        self._dev = {}

    @property
    def m(self) -> int:
        return self.G ** self.D

    def tight_bounds(self):
        return [(lo + 2.01 * (hi - lo) / self.G, hi - 2.01 * (hi - lo) / self.G) for lo, hi in self.bounds]

    def update(self, *Zs, minmax=None) -> bool:
        """gpytorch's rule on the union of the given point sets (or on their per-dimension (mins, maxs) already on the
        host); True when the grid was rebuilt."""
        if minmax is not None:
            mins, maxs = minmax
        else:
            mins = torch.stack([Z.detach().min(0)[0] for Z in Zs]).min(0)[0].double().tolist()
            maxs = torch.stack([Z.detach().max(0)[0] for Z in Zs]).max(0)[0].double().tolist()
        if self.bounds is not None and not any(mn < lo or mx > hi for mn, mx, (lo, hi) in zip(mins, maxs, self.tight_bounds())):
            return False
        sp = [(mx - mn) / (self.G - 4.02) for mn, mx in zip(mins, maxs)]
        sp = [v if v > 0 else 1.0 for v in sp]                  # (a degenerate dimension: any spacing covers it)
        bounds = [(mn - 2.01 * v, mx + 2.01 * v) for mn, mx, v in zip(mins, maxs, sp)]
        if bounds == self.bounds:            # (the tight bounds re-derive min / max with rounding: the same data can land a
            return False                     # few ulps outside them, and gpytorch then rebuilds the IDENTICAL grid)
        self.bounds = bounds
        self.version += 1
        self._dev = {}
        return True

    def nodes_1d(self):
        out = []
        for lo, hi in self.bounds:
            d = (hi - lo) / (self.G - 2)
            out.append((lo - d, (hi - lo + 2 * d) / (self.G - 1)))          # (first node, node spacing)
        return out

    def tensors(self, dtype, device):
        """(g0 [D], inv_delta [D], U [m][D]) on the device; U in node order i0 * G + i1."""
        key = (dtype, str(device))
        if key not in self._dev:
            nd = self.nodes_1d()
            g0 = torch.tensor([a for a, _ in nd], dtype=torch.float64)
            dl = torch.tensor([b for _, b in nd], dtype=torch.float64)
            ax = [g0[d] + dl[d] * torch.arange(self.G, dtype=torch.float64) for d in range(self.D)]
            U = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, self.D)
            self._dev[key] = (g0.to(dtype).to(device), (1.0 / dl).to(dtype).to(device), U.to(dtype).to(device).contiguous())
        return self._dev[key]


def ski_weights(Z: torch.Tensor, grid: SkiGrid, want_cell: bool = False):
    """(base [N][D] int32, w [N][D][4], dw [N][D][4] [, cell [N] int32]) of the points Z on the grid (amx_ski_weights)."""
    Z = Z.detach().contiguous()
    N, D = Z.shape
    g0, inv_delta, _ = grid.tensors(Z.dtype, Z.device)
    base = torch.empty(N, D, dtype=torch.int32, device=Z.device)
    w = torch.empty(N, D, 4, dtype=Z.dtype, device=Z.device)
    dw = torch.empty_like(w)
    cell = torch.empty(N, dtype=torch.int32, device=Z.device) if want_cell else None
    L.call("amx_ski_weights", L.ptr(Z), L.ptr(g0), L.ptr(inv_delta), N, D, grid.G, int(Z.dtype == torch.float64),
           L.ptr(base), L.ptr(w), L.ptr(dw), L.ptr(cell), L.stream_ptr(Z))
    return (base, w, dw, cell) if want_cell else (base, w, dw)


def _ski_cells(base: torch.Tensor, G: int, cell: torch.Tensor = None):
    """Point indices sorted by grid cell and the first position of every cell (plumbing-size integer work in torch)."""
    D = base.shape[1]
    nc = G - 3
    if cell is None:
        cell = base[:, 0] if D == 1 else base[:, 0] * nc + base[:, 1]                # int32
    skey, order = torch.sort(cell, stable=True)
    # first position of every cell in the sorted order (no histogram: torch.bincount synchronises with the host)
    start = torch.searchsorted(skey, torch.arange(nc ** D + 1, dtype=torch.int32, device=base.device), out_int32=True)
    return order.to(torch.int32).contiguous(), start.contiguous()


def ski_gram(base, w, R, grid: SkiGrid, need_A: bool = True, cell=None):
    """A = W^T W [m][m] (or None) and b = W^T r [C][m] for R [C][N] (amx_ski_gram)."""
    N, D = base.shape
    G, m = grid.G, grid.m
    C = 0 if R is None else R.shape[0]
    order, start = _ski_cells(base, G, cell)
    nws = ((G - 3) ** D) * ((4 ** D) ** 2 + C * 4 ** D)
    ws = torch.empty(nws, dtype=w.dtype, device=w.device)
    A = torch.zeros(m, m, dtype=w.dtype, device=w.device) if need_A else None      # (the kernel writes the band only)
    b = torch.empty(C, m, dtype=w.dtype, device=w.device) if C else None
    L.call("amx_ski_gram", L.ptr(w), L.ptr(R.contiguous()) if C else None, L.ptr(order), L.ptr(start), N, D, G, C,
           int(w.dtype == torch.float64), L.ptr(ws), L.ptr(A) if need_A else None, L.ptr(b) if C else None, L.stream_ptr(w))
    return A, b


def ski_interp(base, w, V, grid: SkiGrid):
    """W V^T: [C][N] for node vectors V [C][m] (amx_ski_interp)."""
    N, D = base.shape
    V = V.contiguous()
    Y = torch.empty(V.shape[0], N, dtype=w.dtype, device=w.device)
    L.call("amx_ski_interp", L.ptr(base), L.ptr(w), L.ptr(V), N, D, grid.G, V.shape[0], int(w.dtype == torch.float64),
           L.ptr(Y), L.stream_ptr(w))
    return Y


def ski_cov(base1, w1, base2, w2, Q, grid: SkiGrid, scale: float = 1.0, diag: bool = False):
    """scale * W1 Q W2^T ([N1][N2]) or its diagonal (amx_ski_cov)."""
    N1, D = base1.shape
    N2 = base2.shape[0]
    out = torch.empty(N1 if diag else (N1, N2), dtype=w1.dtype, device=w1.device)
    L.call("amx_ski_cov", L.ptr(base1), L.ptr(w1), N1, L.ptr(base2), L.ptr(w2), N2, L.ptr(Q.contiguous()), D, grid.G,
           float(scale), int(diag), int(w1.dtype == torch.float64), L.ptr(out), L.stream_ptr(w1))
    return out


class _SkiCoreLU:
    """The m x m core shared by training and prediction, for ANY base kernel.  With M = sig2 I + K_UU A (eigenvalues
    sig2 + eig(A^1/2 K_UU A^1/2) >= sig2: LU without a factorisation of the ill-conditioned K_UU itself):
        P = M^-1,  x = P K_UU b = K_UU W^T Khat^-1 r,  log det M,     Khat = W K_UU W^T + sig2 I,
        Q = P K_UU (= posterior covariance of the node values / sig2),   M^-T A,   tr P.
    The LU of the 2500 x 2500 matrix of the default grid is 21 of the 22 ms of this core on the MI355X (rocSOLVER getrf
    through torch: profiles/r06_logs/r06_lu_probe.log) — the reason for _SkiCoreKron below."""

    def __init__(self, U, lengthscale, s2, kind, A, b, sig2, grid, host_ls=None):
        self.A, self.sig2 = A, sig2
        m = A.shape[0]
        with _phase("k_build"):
            Kuu = kernel_matrix(U, U, lengthscale, s2, kind)
        with _phase("grid_solve"):
            M = Kuu @ A
            M.diagonal().add_(sig2)
            LU, piv, _ = torch.linalg.lu_factor_ex(M)      # (no host round trip for an error flag: M is never singular)
            del M
            self.logdet = LU.diagonal().abs().log().sum()
            self.P = torch.linalg.lu_solve(LU, piv, torch.eye(m, dtype=Kuu.dtype, device=Kuu.device))
            self.x = self.P @ (Kuu @ b)
        self._args = (U, lengthscale, s2, kind)

    def Q(self):
        U, ls, s2, kind = self._args
        Qm = self.P @ kernel_matrix(U, U, ls, s2, kind)
        return 0.5 * (Qm + Qm.T)

    def PtA(self):
        T = self.P.T @ self.A
        return 0.5 * (T + T.T)

    def trP(self):
        return torch.diagonal(self.P).sum()


class _SkiCoreKron:
    """The same quantities for the RBF kernel — the only base kernel the reference's dklGPR builds (gp.py:41-44) — without an
    m x m factorisation.  On the product grid the ARD-RBF covariance is a Kronecker product, K_UU = s2 K_0 (x) K_1 with K_d the
    G x G one-dimensional kernels, so its eigen-decomposition costs two G x G symmetric eigenproblems (fp64, on the host:
    0.2 ms each at G = 50): K_UU = V L V^T, V = V_0 (x) V_1, L = s2 l_0 (x) l_1.  A smooth kernel on a fine grid has a tiny
    numerical rank: only the r eigenpairs with L > 1e-16 L_max are kept (r ~ 300 of 2500 at the initial lengthscale; the
    discarded part of K_UU is below the rounding of the kept part).  With F = V_r L_r^1/2 (m x r), K_UU = F F^T and
        T = sig2 I_r + F^T A F  (SPD, r x r),    log det(sig2 I_m + K_UU A) = (m - r) log sig2 + log det T,
        x = F T^-1 F^T b,   Q = F T^-1 F^T,   M^-T A = (A - (A F) T^-1 (A F)^T) / sig2,   tr M^-1 = (m - r) / sig2 + tr T^-1
    (push-through identities of the LU form).  One r x r Cholesky + a handful of m x m x r GEMMs."""

    EPS = 1e-16

    def __init__(self, U, lengthscale, s2, kind, A, b, sig2, grid, host_ls=None):
        assert kind == 0
        self.A, self.sig2 = A, sig2
        dt, dev = A.dtype, A.device
        m, G, D = A.shape[0], grid.G, grid.D
        with _phase("k_build"):
            # host side (numpy, fp64): two G x G eigenproblems and the choice of the kept pairs; the m x r factor itself is
            # formed on the device from the two G x r column selections
            import numpy as np
            ls = host_ls if host_ls is not None else lengthscale.detach().reshape(-1).double().tolist()
            nd = grid.nodes_1d()
            Vs, lams = [], []
            for d in range(D):
                x = (nd[d][1] / ls[d]) * np.arange(G, dtype=np.float64)
                lam, V = np.linalg.eigh(np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2))
                Vs.append(V)
                lams.append(np.maximum(lam, 0.0))
            if D == 1:
                lam = lams[0] * s2
                keep = np.nonzero(lam > self.EPS * lam.max())[0]
                F = torch.from_numpy(Vs[0][:, keep] * np.sqrt(lam[keep])).to(dt).to(dev)
            else:
                lam = (s2 * lams[0][:, None] * lams[1][None, :]).reshape(-1)
                keep = np.nonzero(lam > self.EPS * lam.max())[0]
                a0, a1 = keep // G, keep % G
                V0 = torch.from_numpy(np.ascontiguousarray(Vs[0][:, a0] * np.sqrt(lam[keep]))).to(dev)     # G x r, fp64
                V1 = torch.from_numpy(np.ascontiguousarray(Vs[1][:, a1])).to(dev)
                F = (V0[:, None, :] * V1[None, :, :]).reshape(m, -1).to(dt)
            F = F.contiguous()
        self.r = r = F.shape[1]
        with _phase("grid_solve"):
            self.AF = A @ F
            T = F.T @ self.AF
            T = 0.5 * (T + T.T)
            T.diagonal().add_(sig2)
            Lc, _ = torch.linalg.cholesky_ex(T)
            del T
            self.logdet = (m - r) * math.log(sig2) + 2.0 * torch.log(torch.diagonal(Lc)).sum()
            self.Tinv = torch.cholesky_inverse(Lc)
            self.F = F
            self.x = F @ (self.Tinv @ (F.T @ b))

    def Q(self):
        Qm = (self.F @ self.Tinv) @ self.F.T
        return 0.5 * (Qm + Qm.T)

    def PtA(self):
        T = (self.A - (self.AF @ self.Tinv) @ self.AF.T) / self.sig2
        return 0.5 * (T + T.T)

    def trP(self):
        return (self.A.shape[0] - self.r) / self.sig2 + torch.diagonal(self.Tinv).sum()


def _ski_core(U, lengthscale, s2, kind, A, b, sig2, grid, host_ls=None):
    return (_SkiCoreKron if kind == 0 and SKI_KRON[0] else _SkiCoreLU)(U, lengthscale, s2, kind, A, b, sig2, grid, host_ls)


SKI_KRON = [True]            # (tests compare the two cores)


class _SkiMLLFn(torch.autograd.Function):
    """(Z, Y [q][N], lengthscale [q][D], outputscale [q], noise [q], mean [q]) -> sum over the q outputs of the per-datum
    marginal log likelihood of the KISS-GP model  y ~ N(mean, W K_UU W^T + noise I):
        r^T Khat^-1 r = (r.r - b.x) / noise,          log det Khat = (N - m) log noise + log det(noise I + K_UU A),
    A = W^T W, b = W^T r (csrc/ski.hip), x = (noise I + K_UU A)^-1 K_UU b.  Backward in closed form (u = W^T Khat^-1 r =
    (b - A x) / noise, Q = M^-1 K_UU = noise^-1 * posterior covariance of the grid values):
        d/dA = -(x x^T / noise + Q) / 2,   d/db = x / noise,   d/dK_UU = (u u^T - M^-T A) / 2,
        d/dnoise = (quad - u.x) / (2 noise) - (tr M^-1 + (N - m) / noise) / 2
    then amx_ski_gram_bwd (points) and amx_kernel_matrix_bwd on the grid (lengthscale, outputscale)."""

    @staticmethod
    def forward(ctx, Z, Y, lengthscale, outputscale, noise, mean, kind, grid, host=None):
        # host: (outputscale [q], noise [q], lengthscale [q][D]) as Python floats when the caller has fetched them already
        # (GPRegressionModel.mll brings them over together with the embedding's range in ONE transfer)
        N, D = Z.shape
        q, m = Y.shape[0], grid.m
        if host is None:
            hv = torch.cat([outputscale.detach().reshape(-1), noise.detach().reshape(-1),
                            lengthscale.detach().reshape(-1)]).double().tolist()
            host = (hv[:q], hv[q:2 * q], [hv[2 * q + i * D:2 * q + (i + 1) * D] for i in range(q)])
        with _phase("ski_gram"):
            base, w, dw, cell = ski_weights(Z, grid, want_cell=True)
            R = (Y.detach() - mean.detach().reshape(q, 1)).contiguous()
            A, b = ski_gram(base, w, R, grid, cell=cell)
        _, _, U = grid.tensors(Z.dtype, Z.device)
        tot = 0
        per = []
        for i in range(q):
            sig2, s2 = host[1][i], host[0][i]
            core = _ski_core(U, lengthscale[i], s2, kind, A, b[i], sig2, grid, host[2][i])
            with _phase("grid_solve"):
                x = core.x
                quad = ((R[i] * R[i]).sum() - (b[i] * x).sum()) / sig2
                u = (b[i] - A @ x) / sig2
                tot = tot + (-0.5 * quad - 0.5 * ((N - m) * math.log(sig2) + core.logdet) - 0.5 * N * math.log(2 * math.pi)) / N
            per.append((core, x, u, quad, sig2, s2))
        ctx.save_for_backward(lengthscale.detach())
        ctx.state = (base, w, dw, R, A, per, kind, grid, N, D)
        return tot

    @staticmethod
    def backward(ctx, g):
        (ls,) = ctx.saved_tensors
        base, w, dw, R, A, per, kind, grid, N, D = ctx.state
        q, m = R.shape[0], grid.m
        dt, dev = R.dtype, R.device
        _, _, U = grid.tensors(dt, dev)
        GA = torch.zeros(m, m, dtype=dt, device=dev)
        gb = torch.empty(q, m, dtype=dt, device=dev)
        d_ls, d_s2, d_noise, g_rr = [], [], [], []
        for i, (core, x, u, quad, sig2, s2) in enumerate(per):
            with _phase("grid_bwd"):
                GA.add_(torch.outer(x, x) / sig2 + core.Q(), alpha=-0.5 / N)
                gb[i] = x / (sig2 * N)
                GK = (torch.outer(u, u) - core.PtA()) * (0.5 / N)
                d_noise.append((0.5 * (quad - (u * x).sum()) / sig2 - 0.5 * (core.trP() + (N - m) / sig2)) / N)
                g_rr.append(-0.5 / (sig2 * N))
            with _phase("k_bwd"):
                dU = torch.empty_like(U)
                part = torch.empty((m + 3) // 4, D + 1, dtype=dt, device=dev)
                inv_ls = (1.0 / ls[i].reshape(-1)).to(dt).contiguous()
                L.call("amx_kernel_matrix_bwd", L.ptr(U), L.ptr(inv_ls), s2, kind, m, D, int(dt == torch.float64),
                       L.ptr(GK.contiguous()), L.ptr(dU), L.ptr(part), L.stream_ptr(U))
                tot = part.sum(0)
                d_ls.append((-tot[:D] * inv_ls * inv_ls).reshape(ls[i].shape))
                d_s2.append(tot[D])
                del GK
        with _phase("ski_gram_bwd"):
            dZ = torch.empty(N, D, dtype=dt, device=dev)
            dr = torch.empty(q, N, dtype=dt, device=dev)
            L.call("amx_ski_gram_bwd", L.ptr(base), L.ptr(w), L.ptr(dw), L.ptr(R), L.ptr(GA), L.ptr(gb), N, D, grid.G, q,
                   int(dt == torch.float64), L.ptr(dZ), L.ptr(dr), L.stream_ptr(w))
            dr = dr + 2.0 * torch.tensor(g_rr, dtype=dt, device=dev).reshape(q, 1) * R
        return (g * dZ, g * dr, g * torch.stack(d_ls), g * torch.stack(d_s2), g * torch.stack(d_noise),
                -g * dr.sum(1), None, None, None)


class GPRegressionModel(nn.Module):
    """DKL GP regression module: feature extractor -> ScaleToBounds(-1, 1) -> ConstantMean + ScaleKernel(RBF
    with ARD lengthscales) for each of the q outputs sharing the embedding (gp.py:29-60)."""

    MAX_GRID_NODES = 4096        # dense m x m grid algebra: 50^2 = 2500 nodes = 25 MB per fp32 matrix

    def __init__(self, X: torch.Tensor, y: torch.Tensor, feature_extractor: Type[nn.Module], embedim: int,
                 kernel: str = "rbf", grid_size: int = 50, gp: str = "kissgp") -> None:
        super().__init__()
        q = y.shape[0]
        if gp not in ("kissgp", "exact"):
            raise ValueError("gp must be 'kissgp' (the reference's GridInterpolationKernel model) or 'exact'")
        if gp == "kissgp" and (embedim > 2 or grid_size < 4 or grid_size ** embedim > self.MAX_GRID_NODES or q > 8):
            import warnings
            warnings.warn(f"KISS-GP with grid_size={grid_size}, embedim={embedim}, {q} outputs is outside what the dense grid "
                          "algebra of this build covers (embedim <= 2, grid_size^embedim <= 4096, <= 8 outputs): using the "
                          "exact dense GP instead", UserWarning, stacklevel=2)
            gp = "exact"
        self.gp = gp
        self.grid = SkiGrid(embedim, grid_size) if gp == "kissgp" else None
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
        if self.gp == "kissgp":
            # everything the host needs of this step in ONE device -> host transfer: the embedding's range (gpytorch's grid
            # rule) and the kernel hyper-parameters (the Kronecker core's eigenproblems run on the host)
            q, D = self.train_targets.shape[0], Z.shape[1]
            ls, s2, nz = self.lengthscale, self.outputscale, self.noise[:, 0]
            Zd = Z.detach()
            hv = torch.cat([Zd.min(0)[0].double(), Zd.max(0)[0].double(), s2.detach().reshape(-1).double(),
                            nz.detach().reshape(-1).double(), ls.detach().reshape(-1).double()]).tolist()
            self.grid.update(minmax=(hv[:D], hv[D:2 * D]))
            o = 2 * D
            host = (hv[o:o + q], hv[o + q:o + 2 * q], [hv[o + 2 * q + i * D:o + 2 * q + (i + 1) * D] for i in range(q)])
            return _SkiMLLFn.apply(Z, self.train_targets, ls, s2, nz, self.mean_constant[:, 0], self.kind, self.grid, host)
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
        key = self._state_key() + ((self.grid.version,) if self.gp == "kissgp" else ())
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1], self._cache[2]
        Z = self.embed(self.train_inputs[0])
        factors = []
        if self.gp == "kissgp":
            # per output: x = K_UU W^T Khat^-1 r (the node values of the predictive mean) and noise * Q = the posterior
            # covariance of the node values, both m-sized; the same core as the training step (_ski_core)
            grid = self.grid
            grid.update(Z)                   # (gpytorch applies its grid rule on every kernel call, train or eval mode)
            base, w, _ = ski_weights(Z, grid)
            q = self.train_targets.shape[0]
            R = (self.train_targets - self.mean_constant.reshape(q, 1)).contiguous()
            A, b = ski_gram(base, w, R, grid)
            _, _, U = grid.tensors(Z.dtype, Z.device)
            for i in range(q):
                nz = float(self.noise[i, 0])
                core = _ski_core(U, self.lengthscale[i], float(self.outputscale[i]), self.kind, A, b[i], nz, grid)
                factors.append((core.x, nz * core.Q()))
            self._cache = (self._state_key() + (grid.version,), Z, factors)
            self.n_factorisations = getattr(self, "n_factorisations", 0) + 1
            return Z, factors
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
        if self.gp == "kissgp":
            # gpytorch evaluates the kernel on cat(train, test) at prediction time: test points outside the tight bounds
            # rebuild the (dynamic) grid over the union of both sets
            if self.grid.update(Z, Zs):
                Z, factors = self._posterior_factors()
            bs, ws, _ = ski_weights(Zs, self.grid)
            for i, (x, Qs) in enumerate(factors):
                means.append(self.mean_constant[i, 0] + ski_interp(bs, ws, x.reshape(1, -1), self.grid).reshape(-1))
                v = ski_cov(bs, ws, bs, ws, Qs, self.grid, 1.0, diag=not full_cov)
                vars_.append(v if full_cov else v.clamp_min(0))
            return torch.stack(means), torch.stack(vars_)
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
