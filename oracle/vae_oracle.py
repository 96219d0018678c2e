"""CPU ORACLE for the VAE / rVAE hot path (encoder, spatial decoder, ELBO).

TEST INFRASTRUCTURE ONLY — never imported by the product package.  Functional restatement through
stock PyTorch CPU ops of what the reference's modules compute; each function cites the reference
lines it follows.  Gradients via torch autograd over these ops.

Parity status: PINNED against tests/golden/vae.npz (generated from the real reference with the
reparameterisation noise injected, fp32 and fp64; oracle/make_golden.py, tests/test_oracle_vae_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


def imcoordgrid(im_dim: Tuple[int, int], dtype=torch.float32) -> Tensor:
    """(H*W, 2) pixel coordinates, x in linspace(-1,1,H) (rows), y in linspace(1,-1,W) (cols), 'ij' order
    (atomai/utils/coords.py:37-54)."""
    xx = torch.linspace(-1, 1, im_dim[0])
    yy = torch.linspace(1, -1, im_dim[1])
    x0, x1 = torch.meshgrid(xx, yy, indexing="ij")
    return torch.stack((x0.reshape(-1), x1.reshape(-1)), 1).to(dtype)


def transform_coordinates(coord: Tensor, phi: Tensor, coord_dx=0) -> Tensor:
    """coord @ [[cos, sin], [-sin, cos]] + dx, batched (atomai/utils/coords.py:57-83)."""
    c, s = torch.cos(phi)[:, None], torch.sin(phi)[:, None]
    x, y = coord[..., 0], coord[..., 1]
    out = torch.stack((x * c - y * s, x * s + y * c), -1)
    return out + coord_dx


def fc_encoder(sd: StateDict, x: Tensor, num_layers: int = 2, softplus_out: bool = False):
    """fcEncoderNet.forward (atomai/nets/ed.py:334-343)."""
    h = x.reshape(x.shape[0], -1)
    for i in range(num_layers):
        h = torch.tanh(torch.nn.functional.linear(h, sd[f"dense.{2 * i}.weight"], sd[f"dense.{2 * i}.bias"]))
    mu = torch.nn.functional.linear(h, sd["fc11.weight"], sd["fc11.bias"])
    ls = torch.nn.functional.linear(h, sd["fc12.weight"], sd["fc12.bias"])
    if softplus_out:
        ls = torch.nn.functional.softplus(ls)
    return mu, ls


def conv_encoder(sd: StateDict, x: Tensor, num_layers: int = 2, lrelu_a: float = 0.1,
                 softplus_out: bool = False):
    """convEncoderNet.forward (atomai/nets/ed.py:276-289): ConvBlock without BatchNorm
    (conv 3x3 pad 1 -> LeakyReLU(lrelu_a), atomai/nets/blocks.py:59-76; Sequential indices 0,2,4..),
    NCHW flatten, two Linear heads."""
    F = torch.nn.functional
    h = x.unsqueeze(1) if x.ndim in (2, 3) else x.permute(0, 3, 1, 2)
    for i in range(num_layers):
        h = F.leaky_relu(F.conv2d(h, sd[f"conv.block.{2 * i}.weight"], sd[f"conv.block.{2 * i}.bias"],
                                  padding=1), lrelu_a)
    h = h.reshape(h.shape[0], -1)
    mu = F.linear(h, sd["fc11.weight"], sd["fc11.bias"])
    ls = F.linear(h, sd["fc12.weight"], sd["fc12.bias"])
    if softplus_out:
        ls = F.softplus(ls)
    return mu, ls


def r_decoder(sd: StateDict, x_coord: Tensor, z: Tensor, out_hw: Tuple[int, int], num_layers: int = 2,
              skip: bool = False) -> Tensor:
    """rDecoderNet.forward + coord_latent.forward (atomai/nets/ed.py:626-642, 672-687)."""
    B, n = x_coord.shape[:2]
    F = torch.nn.functional
    h = F.linear(x_coord.reshape(B * n, -1), sd["coord_latent.fc_coord.weight"], sd["coord_latent.fc_coord.bias"])
    h = h.reshape(B, n, -1) + F.linear(z, sd["coord_latent.fc_latent.weight"])[:, None]
    h = h.reshape(B * n, -1)
    if not skip:
        h = torch.tanh(h)
    res = h
    for i in range(num_layers):
        h = torch.tanh(F.linear(h, sd[f"fc_decoder.{2 * i}.weight"], sd[f"fc_decoder.{2 * i}.bias"]))
        if skip:
            h = h + res
    h = F.linear(h, sd["out.weight"], sd["out.bias"])
    return h.reshape(B, *out_hw)


def fc_decoder(sd: StateDict, z: Tensor, out_hw: Tuple[int, int], num_layers: int = 2) -> Tensor:
    """fcDecoderNet.forward for single-channel images (atomai/nets/ed.py:569-580)."""
    F = torch.nn.functional
    h = z
    for i in range(num_layers):
        h = torch.tanh(F.linear(h, sd[f"decoder.{2 * i}.weight"], sd[f"decoder.{2 * i}.bias"]))
    h = F.linear(h, sd["out.weight"], sd["out.bias"])
    return h.reshape(-1, *out_hw)


def reconstruction_mse(x: Tensor, x_rec: Tensor) -> Tensor:
    """0.5 * sum over pixels of (x_rec - x)^2, per sample (losses_metrics/vi_losses.py:23-26)."""
    B = x.shape[0]
    return 0.5 * ((x_rec.reshape(B, -1) - x.reshape(B, -1)) ** 2).sum(1)


def reconstruction_ce(x: Tensor, x_rec: Tensor, in_dim: Sequence[int]) -> Tensor:
    """'ce' branch (vi_losses.py:27-34): binary_cross_entropy_with_logits(reduction='none') of the decoder output
    against the image, reshaped to (-1, H*W) for a 2-D in_dim and to (-1, H*W, C) for a 3-D one, then ``.sum(-1)``:
    per-SAMPLE sums (B,) in the first case, per-PIXEL sums over the channels (B, H*W) in the second — the callers'
    ``.mean()`` then averages over samples, or over samples x pixels."""
    rs = (int(in_dim[0]) * int(in_dim[1]),)
    if len(in_dim) == 3:
        rs = rs + (int(in_dim[-1]),)
    return torch.nn.functional.binary_cross_entropy_with_logits(x_rec.reshape(-1, *rs), x.reshape(-1, *rs),
                                                                reduction="none").sum(-1)


def reconstruction(loss: str, x: Tensor, x_rec: Tensor, in_dim=None) -> Tensor:
    if loss == "mse":
        return reconstruction_mse(x, x_rec)
    if loss == "ce":
        return reconstruction_ce(x, x_rec, in_dim if in_dim is not None else tuple(x.shape[1:]))
    raise NotImplementedError("Reconstruction loss must be 'mse' or 'ce'")


def kld_normal(mu: Tensor, log_sd: Tensor) -> Tensor:
    """KL(N(mu, sd) || N(0,1)) summed over latent dims (vi_losses.py:40-57)."""
    return (-log_sd + 0.5 * torch.exp(log_sd) ** 2 + 0.5 * mu ** 2 - 0.5).sum(-1)


def kld_rot(phi_prior: float, phi_logsd: Tensor) -> Tensor:
    """vi_losses.py:77-84 (np.log(phi_prior) is a python float there)."""
    return -phi_logsd + math.log(phi_prior) + torch.exp(phi_logsd) ** 2 / (2 * phi_prior ** 2) - 0.5


def infocapacity(kl: Tensor, capacity: Sequence[float], num_iter: int) -> Tensor:
    """gamma * |KL - C(num_iter)| (vi_losses.py:224-236)."""
    cmax, niter, gamma = capacity
    cap = min(cmax * (num_iter / float(niter)), cmax)
    return gamma * torch.abs(kl - cap)


def rvae_elbo(x, x_rec, z_mean, z_logsd, phi_prior: float = 0.1, capacity=None, num_iter: int = 0,
              loss: str = "mse", in_dim=None) -> Tensor:
    """rvae_loss (vi_losses.py:111-137): the rotation latent gets the dedicated prior, the
    translation latents enter kld_normal together with the content latents."""
    like = -reconstruction(loss, x, x_rec, in_dim).mean()
    kl = kld_normal(z_mean[:, 1:], z_logsd[:, 1:]).mean() + kld_rot(phi_prior, z_logsd[:, 0]).mean()
    if capacity is not None:
        kl = infocapacity(kl, capacity, num_iter)
    return like - kl


def vae_elbo(x, x_rec, z_mean, z_logsd, capacity=None, num_iter: int = 0, loss: str = "mse", in_dim=None) -> Tensor:
    """vae_loss (vi_losses.py:87-108)."""
    like = -reconstruction(loss, x, x_rec, in_dim).mean()
    kl = kld_normal(z_mean, z_logsd).mean()
    if capacity is not None:
        kl = infocapacity(kl, capacity, num_iter)
    return like - kl


def rvae_forward_elbo(enc: StateDict, dec: StateDict, x: Tensor, eps: Tensor, x_coord: Tensor,
                      translation: bool = True, dx_prior: float = 0.1, phi_prior: float = 0.1,
                      skip: bool = False, capacity=None, num_iter: int = 0, num_layers=(2, 2),
                      conv_enc: bool = False, loss: str = "mse") -> Tensor:
    """rVAE.forward_compute_elbo, training mode, with the reparameterisation noise injected
    (atomai/models/dgm/rvae.py:110-147; trainers/vitrainer.py:223-234)."""
    B = x.shape[0]
    z_mean, z_logsd = (conv_encoder if conv_enc else fc_encoder)(enc, x, num_layers[0])
    z = z_mean + torch.exp(z_logsd) * eps[:, : z_mean.shape[1]]
    phi = z[:, 0]
    if translation:
        dx = (z[:, 1:3] * dx_prior).unsqueeze(1)
        zc = z[:, 3:]
    else:
        dx, zc = 0, z[:, 1:]
    coords = transform_coordinates(x_coord.expand(B, *x_coord.shape), phi, dx)
    x_rec = r_decoder(dec, coords, zc, tuple(x.shape[1:]), num_layers[1], skip)     # (H, W) or (H, W, C), ed.py:606-609
    return rvae_elbo(x, x_rec, z_mean, z_logsd, phi_prior, capacity, num_iter, loss, tuple(x.shape[1:]))


def vae_forward_elbo(enc: StateDict, dec: StateDict, x: Tensor, eps: Tensor, capacity=None,
                     num_iter: int = 0, num_layers=(2, 2), loss: str = "mse") -> Tensor:
    """VAE.forward_compute_elbo (atomai/models/dgm/vae.py:661-687)."""
    z_mean, z_logsd = fc_encoder(enc, x, num_layers[0])
    z = z_mean + torch.exp(z_logsd) * eps[:, : z_mean.shape[1]]
    x_rec = fc_decoder(dec, z, tuple(x.shape[1:3]), num_layers[1])
    return vae_elbo(x, x_rec, z_mean, z_logsd, capacity, num_iter, loss, tuple(x.shape[1:]))
