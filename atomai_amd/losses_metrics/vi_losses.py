"""ELBO objectives of VAE / rVAE on the HIP path (reference: atomai/losses_metrics/vi_losses.py:13-137, 224-236).

Same function names, signatures and semantics (incl. the quirk that the translation latents enter the
plain KL term together with the content latents).  The three per-sample terms — reconstruction, KL(z),
KL(rotation) — come from ONE kernel (csrc/elbo.hip, forward and backward); only their means and the optional
capacity term are combined here, on B-element vectors.
"""
from typing import List, Tuple, Union

import torch

from .. import _lib as L


def _recon_kind(recon_loss: str, in_dim) -> Tuple[int, float]:
    """(recon_kind, recon_scale) of csrc/elbo.hip for reconstruction_loss(recon_loss, in_dim, ...) (vi_losses.py:13-37):
    'mse' -> (0, 1); 'ce' -> (1, 1) for a 2-D in_dim and (1, 1 / (H * W)) for a 3-D one, where the reference's reshape to
    (-1, H*W, C) makes ``.sum(-1)`` a sum over the channels and the ELBO's ``.mean()`` an average over samples x pixels."""
    if recon_loss == "mse":
        return 0, 1.0
    if recon_loss == "ce":
        return 1, (1.0 / (int(in_dim[0]) * int(in_dim[1])) if len(in_dim) == 3 else 1.0)
    raise NotImplementedError("Reconstruction loss must be 'mse' or 'ce'")


class _ElboTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float, kind: int = 0, rscale: float = 1.0):
        B = x.shape[0]
        xf = x.detach().reshape(B, -1).contiguous().float()
        xr = x_rec.detach().reshape(B, -1).contiguous()
        zm, zl = z_mean.detach().contiguous(), z_logsd.detach().contiguous()
        n, Z = xf.shape[1], zm.shape[1]
        recon, klz, klrot = (torch.empty(B, dtype=torch.float32, device=xf.device) for _ in range(3))
        L.call("amx_elbo_terms_fwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), B, n, Z, int(rot),
               float(phi_prior), int(kind), float(rscale), L.ptr(recon), L.ptr(klz), L.ptr(klrot), L.stream_ptr(xf))
        ctx.save_for_backward(xf, xr, zm, zl)
        ctx.meta = (rot, float(phi_prior), x_rec.shape, int(kind), float(rscale))
        return recon, klz, klrot

    @staticmethod
    def backward(ctx, g_recon, g_klz, g_klrot):
        xf, xr, zm, zl = ctx.saved_tensors
        rot, phi_prior, shape, kind, rscale = ctx.meta
        B, n, Z = xf.shape[0], xf.shape[1], zm.shape[1]
        dx, dm, dl = torch.empty_like(xr), torch.empty_like(zm), torch.empty_like(zl)
        z = lambda g: (g if g is not None else torch.zeros(B, device=xf.device)).contiguous().float()
        L.call("amx_elbo_terms_bwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), L.ptr(z(g_recon)),
               L.ptr(z(g_klz)), L.ptr(z(g_klrot)), B, n, Z, int(rot), phi_prior, kind, rscale, L.ptr(dx), L.ptr(dm),
               L.ptr(dl), L.stream_ptr(xf))
        return None, dx.view(shape), dm, dl, None, None, None, None


class _ElboScalarFn(torch.autograd.Function):
    """ELBO = -mean(recon) - mean(klz) [- mean(klrot)] as ONE scalar (no capacity term): the per-sample kernel, a
    one-block combine, and a backward that reads the upstream scalar from device memory (csrc/elbo.hip)."""

    @staticmethod
    def forward(ctx, x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float, kind: int = 0, rscale: float = 1.0):
        B = x.shape[0]
        xf = x.detach().reshape(B, -1).contiguous().float()
        xr = x_rec.detach().reshape(B, -1).contiguous()
        zm, zl = z_mean.detach().contiguous(), z_logsd.detach().contiguous()
        n, Z = xf.shape[1], zm.shape[1]
        terms = torch.empty(3, B, dtype=torch.float32, device=xf.device)
        sp = L.stream_ptr(xf)
        L.call("amx_elbo_terms_fwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), B, n, Z, int(rot), float(phi_prior),
               int(kind), float(rscale), L.ptr(terms[0]), L.ptr(terms[1]), L.ptr(terms[2]), sp)
        out = torch.empty((), dtype=torch.float32, device=xf.device)
        L.call("amx_elbo_combine", L.ptr(terms[0]), L.ptr(terms[1]), L.ptr(terms[2]) if rot else None, B, L.ptr(out), sp)
        ctx.save_for_backward(xf, xr, zm, zl)
        ctx.meta = (rot, float(phi_prior), x_rec.shape, int(kind), float(rscale))
        return out

    @staticmethod
    def backward(ctx, g):
        xf, xr, zm, zl = ctx.saved_tensors
        rot, phi_prior, shape, kind, rscale = ctx.meta
        B, n, Z = xf.shape[0], xf.shape[1], zm.shape[1]
        dx, dm, dl = torch.empty_like(xr), torch.empty_like(zm), torch.empty_like(zl)
        gs = g.detach().reshape(1).float().contiguous()
        L.call("amx_elbo_bwd_scalar", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), L.ptr(gs), -1.0 / B, B, n, Z, int(rot),
               phi_prior, kind, rscale, L.ptr(dx), L.ptr(dm), L.ptr(dl), L.stream_ptr(xf))
        return None, dx.view(shape), dm, dl, None, None, None, None


def elbo_terms(x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float = 0.1, kind: int = 0, rscale: float = 1.0):
    """Per-sample (reconstruction, KL(z), KL(rotation)) vectors; (kind, rscale) from ``_recon_kind``."""
    return _ElboTermsFn.apply(x, x_rec, z_mean, z_logsd, rot, phi_prior, kind, rscale)


def infocapacity(kl_cont_loss: torch.Tensor, cont_capacity: List[float], num_iter: int = 0) -> torch.Tensor:
    """gamma * |KL - C(num_iter)| (vi_losses.py:224-236, continuous channel)."""
    cont_max, cont_num_iters, cont_gamma = cont_capacity
    cont_cap = min(cont_max * (num_iter / float(cont_num_iters)), cont_max)
    return cont_gamma * torch.abs(kl_cont_loss - cont_cap)


def _check(recon_loss, args):
    if len(args) != 2:
        raise ValueError("Pass mean and SD values of encoded distribution as args")
    return args


def vae_loss(recon_loss: str, in_dim: Tuple[int], x: torch.Tensor, x_reconstr: torch.Tensor,
             *args: torch.Tensor, **kwargs: List[float]) -> torch.Tensor:
    """ELBO of a plain VAE (vi_losses.py:87-108)."""
    z_mean, z_logsd = _check(recon_loss, args)
    kind, rscale = _recon_kind(recon_loss, in_dim)
    if kwargs.get("capacity") is None and x.is_cuda | L.is_test_backend():
        return _ElboScalarFn.apply(x, x_reconstr, z_mean, z_logsd, False, 0.1, kind, rscale)
    recon, klz, _ = elbo_terms(x, x_reconstr, z_mean, z_logsd, False, 0.1, kind, rscale)
    kl_div = klz.mean()
    if kwargs.get("capacity") is not None:
        kl_div = infocapacity(kl_div, kwargs["capacity"], num_iter=kwargs.get("num_iter", 0))
    return -recon.mean() - kl_div


def rvae_loss(recon_loss: str, in_dim: Tuple[int], x: torch.Tensor, x_reconstr: torch.Tensor,
              *args: torch.Tensor, **kwargs: Union[List[float], float]) -> torch.Tensor:
    """ELBO of the rotationally invariant VAE (vi_losses.py:111-137)."""
    z_mean, z_logsd = _check(recon_loss, args)
    kind, rscale = _recon_kind(recon_loss, in_dim)
    if kwargs.get("capacity") is None and x.is_cuda | L.is_test_backend():
        return _ElboScalarFn.apply(x, x_reconstr, z_mean, z_logsd, True, kwargs.get("phi_prior", 0.1), kind, rscale)
    recon, klz, klrot = elbo_terms(x, x_reconstr, z_mean, z_logsd, True, kwargs.get("phi_prior", 0.1), kind, rscale)
    kl_div = klz.mean() + klrot.mean()
    if kwargs.get("capacity") is not None:
        kl_div = infocapacity(kl_div, kwargs["capacity"], num_iter=kwargs.get("num_iter", 0))
    return -recon.mean() - kl_div
