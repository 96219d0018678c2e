"""ELBO objectives of VAE / rVAE on the HIP path (reference: atomai/losses_metrics/vi_losses.py:13-137, 224-236).

Same function names, signatures and semantics (incl. the quirk that the translation latents enter the
plain KL term together with the content latents).  The three per-sample terms — reconstruction, KL(z),
KL(rotation) — come from ONE kernel (csrc/elbo.hip, forward and backward); only their means and the optional
capacity term are combined here, on B-element vectors.
"""
from typing import List, Tuple, Union

import torch

from .. import _lib as L


class _ElboTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float):
        B = x.shape[0]
        xf = x.detach().reshape(B, -1).contiguous().float()
        xr = x_rec.detach().reshape(B, -1).contiguous()
        zm, zl = z_mean.detach().contiguous(), z_logsd.detach().contiguous()
        n, Z = xf.shape[1], zm.shape[1]
        recon, klz, klrot = (torch.empty(B, dtype=torch.float32, device=xf.device) for _ in range(3))
        L.call("amx_elbo_terms_fwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), B, n, Z, int(rot),
               float(phi_prior), L.ptr(recon), L.ptr(klz), L.ptr(klrot), L.stream_ptr(xf))
        ctx.save_for_backward(xf, xr, zm, zl)
        ctx.meta = (rot, float(phi_prior), x_rec.shape)
        return recon, klz, klrot

    @staticmethod
    def backward(ctx, g_recon, g_klz, g_klrot):
        xf, xr, zm, zl = ctx.saved_tensors
        rot, phi_prior, shape = ctx.meta
        B, n, Z = xf.shape[0], xf.shape[1], zm.shape[1]
        dx, dm, dl = torch.empty_like(xr), torch.empty_like(zm), torch.empty_like(zl)
        z = lambda g: (g if g is not None else torch.zeros(B, device=xf.device)).contiguous().float()
        L.call("amx_elbo_terms_bwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), L.ptr(z(g_recon)),
               L.ptr(z(g_klz)), L.ptr(z(g_klrot)), B, n, Z, int(rot), phi_prior, L.ptr(dx), L.ptr(dm),
               L.ptr(dl), L.stream_ptr(xf))
        return None, dx.view(shape), dm, dl, None, None


class _ElboScalarFn(torch.autograd.Function):
    """ELBO = -mean(recon) - mean(klz) [- mean(klrot)] as ONE scalar (no capacity term): the per-sample kernel, a
    one-block combine, and a backward that reads the upstream scalar from device memory (csrc/elbo.hip)."""

    @staticmethod
    def forward(ctx, x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float):
        B = x.shape[0]
        xf = x.detach().reshape(B, -1).contiguous().float()
        xr = x_rec.detach().reshape(B, -1).contiguous()
        zm, zl = z_mean.detach().contiguous(), z_logsd.detach().contiguous()
        n, Z = xf.shape[1], zm.shape[1]
        terms = torch.empty(3, B, dtype=torch.float32, device=xf.device)
        sp = L.stream_ptr(xf)
        L.call("amx_elbo_terms_fwd", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), B, n, Z, int(rot), float(phi_prior),
               L.ptr(terms[0]), L.ptr(terms[1]), L.ptr(terms[2]), sp)
        out = torch.empty((), dtype=torch.float32, device=xf.device)
        L.call("amx_elbo_combine", L.ptr(terms[0]), L.ptr(terms[1]), L.ptr(terms[2]) if rot else None, B, L.ptr(out), sp)
        ctx.save_for_backward(xf, xr, zm, zl)
        ctx.meta = (rot, float(phi_prior), x_rec.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        xf, xr, zm, zl = ctx.saved_tensors
        rot, phi_prior, shape = ctx.meta
        B, n, Z = xf.shape[0], xf.shape[1], zm.shape[1]
        dx, dm, dl = torch.empty_like(xr), torch.empty_like(zm), torch.empty_like(zl)
        gs = g.detach().reshape(1).float().contiguous()
        L.call("amx_elbo_bwd_scalar", L.ptr(xf), L.ptr(xr), L.ptr(zm), L.ptr(zl), L.ptr(gs), -1.0 / B, B, n, Z, int(rot),
               phi_prior, L.ptr(dx), L.ptr(dm), L.ptr(dl), L.stream_ptr(xf))
        return None, dx.view(shape), dm, dl, None, None


def elbo_terms(x, x_rec, z_mean, z_logsd, rot: bool, phi_prior: float = 0.1):
    """Per-sample (reconstruction 'mse', KL(z), KL(rotation)) vectors."""
    return _ElboTermsFn.apply(x, x_rec, z_mean, z_logsd, rot, phi_prior)


def infocapacity(kl_cont_loss: torch.Tensor, cont_capacity: List[float], num_iter: int = 0) -> torch.Tensor:
    """gamma * |KL - C(num_iter)| (vi_losses.py:224-236, continuous channel)."""
    cont_max, cont_num_iters, cont_gamma = cont_capacity
    cont_cap = min(cont_max * (num_iter / float(cont_num_iters)), cont_max)
    return cont_gamma * torch.abs(kl_cont_loss - cont_cap)


def _check(recon_loss, args):
    if len(args) != 2:
        raise ValueError("Pass mean and SD values of encoded distribution as args")
    if recon_loss != "mse":
        raise NotImplementedError("Reconstruction loss 'ce' is outside this build's hot path ('mse' only)")
    return args


def vae_loss(recon_loss: str, in_dim: Tuple[int], x: torch.Tensor, x_reconstr: torch.Tensor,
             *args: torch.Tensor, **kwargs: List[float]) -> torch.Tensor:
    """ELBO of a plain VAE (vi_losses.py:87-108)."""
    z_mean, z_logsd = _check(recon_loss, args)
    if kwargs.get("capacity") is None and x.is_cuda | L.is_test_backend():
        return _ElboScalarFn.apply(x, x_reconstr, z_mean, z_logsd, False, 0.1)
    recon, klz, _ = elbo_terms(x, x_reconstr, z_mean, z_logsd, False)
    kl_div = klz.mean()
    if kwargs.get("capacity") is not None:
        kl_div = infocapacity(kl_div, kwargs["capacity"], num_iter=kwargs.get("num_iter", 0))
    return -recon.mean() - kl_div


def rvae_loss(recon_loss: str, in_dim: Tuple[int], x: torch.Tensor, x_reconstr: torch.Tensor,
              *args: torch.Tensor, **kwargs: Union[List[float], float]) -> torch.Tensor:
    """ELBO of the rotationally invariant VAE (vi_losses.py:111-137)."""
    z_mean, z_logsd = _check(recon_loss, args)
    if kwargs.get("capacity") is None and x.is_cuda | L.is_test_backend():
        return _ElboScalarFn.apply(x, x_reconstr, z_mean, z_logsd, True, kwargs.get("phi_prior", 0.1))
    recon, klz, klrot = elbo_terms(x, x_reconstr, z_mean, z_logsd, True, kwargs.get("phi_prior", 0.1))
    kl_div = klz.mean() + klrot.mean()
    if kwargs.get("capacity") is not None:
        kl_div = infocapacity(kl_div, kwargs["capacity"], num_iter=kwargs.get("num_iter", 0))
    return -recon.mean() - kl_div
