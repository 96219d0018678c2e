"""dklGPR: deep kernel learning GP regression (reference: atomai/models/dklgp/dklgpr.py:23-241)."""
from typing import Tuple

import numpy as np
import torch

from ...trainers.gptrainer import dklGPTrainer


class dklGPR(dklGPTrainer):
    """``dklGPR(indim, embedim=2, shared_embedding_space=True, precision='double', ...)``"""

    _noted_gp = False

    def __init__(self, indim: int, embedim: int = 2, shared_embedding_space: bool = True, **kwargs) -> None:
        super().__init__(indim, embedim, shared_embedding_space, **kwargs)
        if not dklGPR._noted_gp:
            dklGPR._noted_gp = True
            # Stated once per process ("parity unpinned")
            import warnings
            warnings.warn("atomai_amd.dklGPR: the GP layer is the reference's KISS-GP model (GridInterpolationKernel, "
                          "grid_size=50 - atomai/nets/gp.py:41-46) with its marginal likelihood and posterior evaluated "
                          "EXACTLY on the grid (csrc/ski.hip + dense m x m algebra), where gpytorch uses CG / Lanczos / "
                          "LOVE estimators: results agree with the reference up to gpytorch's solver tolerances; embedim > 2 "
                          "and gp='exact' run a dense exact GP instead.  There is no gpytorch in this build to pin either "
                          "against (oracle/gp_oracle.py)", UserWarning, stacklevel=2)

    def fit(self, X, y, training_cycles: int = 1, **kwargs) -> None:
        _ = self.run(X, y, training_cycles, **kwargs)

    def fit_ensemble(self, X, y, training_cycles: int = 1, n_models: int = 5, **kwargs) -> None:
        """Trains ``n_models`` independently initialised DKL-GP models on the same scalar target (dklgpr.py:95-131)."""
        import warnings
        if y.ndim == 1:
            y = y[None]
        if y.shape[0] > 1:
            raise NotImplementedError("The ensemble training is currently supported only for scalar targets")
        y = y.repeat(n_models, 0) if isinstance(y, np.ndarray) else y.repeat(n_models, 1)
        if self.correlated_output:
            warnings.warn("Replacing a single shared embedding space with {} independent ones".format(n_models))
            self.correlated_output = False
        self.ensemble = True
        _ = self.run(X, y, training_cycles, **kwargs)

    def _compute_posterior(self, X: torch.Tensor, full_cov: bool = False):
        self.gp_model.eval()
        return self.gp_model.posterior(X.to(self.device), full_cov)

    def _draw(self, X, num_samples):
        mean, cov = self._compute_posterior(X, full_cov=True)
        n = cov.shape[-1]
        cov = 0.5 * (cov + cov.transpose(-1, -2))
        eye = torch.eye(n, dtype=cov.dtype, device=cov.device)
        base = cov.diagonal(dim1=-2, dim2=-1).mean(-1).clamp_min(1e-12)
        Lc, scale = None, 1e-6 if cov.dtype == torch.float32 else 1e-8
        for _ in range(7):                    # growing diagonal jitter, as gpytorch's psd_safe_cholesky does
            Lc, info = torch.linalg.cholesky_ex(cov + (scale * base)[:, None, None] * eye)
            if int(info.abs().max()) == 0:
                break
            scale *= 10
        else:
            raise RuntimeError("posterior covariance is not positive definite even with a 1e0 relative jitter")
        eps = torch.randn(num_samples, cov.shape[0], n, 1, dtype=cov.dtype, device=cov.device)
        return mean[None] + (Lc[None] @ eps).squeeze(-1)

    def sample_from_posterior(self, X, num_samples: int = 1000) -> np.ndarray:
        X, _ = self.set_data(X)
        return self._draw(X, num_samples).cpu().numpy()

    def thompson(self, X_cand, scalarize_func=None, maximize: bool = True) -> Tuple[np.ndarray, int]:
        X_cand, _ = self.set_data(X_cand)
        tsample = self._draw(X_cand, 1)[0]
        if tsample.ndim > 1 and scalarize_func is not None:
            tsample = scalarize_func(tsample).unsqueeze(0)
        idx = tsample.argmax(1) if maximize else tsample.argmin(1)
        return tsample.cpu().numpy(), idx.cpu().numpy()

    def _predict(self, x_new: torch.Tensor) -> Tuple[torch.Tensor]:
        mean, var = self._compute_posterior(x_new)
        return mean.cpu(), var.cpu()

    def predict(self, x_new, **kwargs) -> Tuple[np.ndarray]:
        """Posterior mean and variance in ``batch_size`` chunks (dklgpr.py:202-217)."""
        x_new, _ = self.set_data(x_new, device='cpu')
        bs = kwargs.get("batch_size", len(x_new))
        means, vars_ = [], []
        for i in range(0, len(x_new), bs):
            m, v = self._predict(x_new[i:i + bs])
            means.append(m)
            vars_.append(v)
        return torch.cat(means, 1).numpy().squeeze(), torch.cat(vars_, 1).numpy().squeeze()

    def _embed(self, x_new: torch.Tensor):
        self.gp_model.eval()
        with torch.no_grad():
            return self.gp_model.embed(x_new.to(self.device)).cpu()

    def embed(self, x_new, **kwargs) -> torch.Tensor:
        """Embeds the input data to the latent space of the (trained) feature extractor (dklgpr.py:231-241)."""
        x_new, _ = self.set_data(x_new, device='cpu')
        bs = kwargs.get("batch_size", len(x_new))
        out = [self._embed(x_new[i:i + bs]) for i in range(0, len(x_new), bs)]
        # independent per-output networks (GPModelList) embed to (q, n, embedim): the chunks join along the SAMPLE axis
        return torch.cat(out, dim=-2).numpy()
