"""rVAE: rotationally (and translationally) invariant VAE (reference: atomai/models/dgm/rvae.py:22-219)."""
from copy import deepcopy as dc
from typing import Optional

import torch

from ...losses_metrics import rvae_loss
from ...utils import set_train_rng, to_onehot
from .vae import BaseVAE


class _LatentFn(torch.autograd.Function):
    """(z_mean, z_logsd, eps) -> (theta (B, 3), content latents): reparameterisation, the (phi, dx, dy) split, the
    translation prior and the concatenation of rvae.py:118-137 in one kernel each way (csrc/elbo.hip)."""

    @staticmethod
    def forward(ctx, z_mean, z_logsd, eps, translation: bool, dx_prior: float):
        from ... import _lib as L
        zm, zl, ep = z_mean.detach().contiguous(), z_logsd.detach().contiguous(), eps.detach().contiguous()
        B, Z = zm.shape
        skip = 3 if translation else 1
        theta = torch.empty(B, 3, dtype=torch.float32, device=zm.device)
        zc = torch.empty(B, Z - skip, dtype=torch.float32, device=zm.device)
        L.call("amx_rvae_latent_fwd", L.ptr(zm), L.ptr(zl), L.ptr(ep), B, Z, int(translation), float(dx_prior),
               L.ptr(theta), L.ptr(zc) if Z > skip else None, L.stream_ptr(zm))
        ctx.save_for_backward(zl, ep)
        ctx.meta = (translation, float(dx_prior))
        return theta, zc

    @staticmethod
    def backward(ctx, dtheta, dzc):
        from ... import _lib as L
        zl, ep = ctx.saved_tensors
        translation, dx_prior = ctx.meta
        B, Z = zl.shape
        dm, dl = torch.empty_like(zl), torch.empty_like(zl)
        dt = None if dtheta is None else dtheta.contiguous()
        dc = None if (dzc is None or dzc.numel() == 0) else dzc.contiguous()
        L.call("amx_rvae_latent_bwd", L.ptr(zl), L.ptr(ep), L.ptr(dt), L.ptr(dc), B, Z, int(translation), dx_prior,
               L.ptr(dm), L.ptr(dl), L.stream_ptr(zl))
        return dm, dl, None, None, None


class rVAE(BaseVAE):
    """``rVAE(in_dim, latent_dim=2, translation=True, seed=0, **kwargs)`` with the spatial decoder on the fused
    HIP kernels.  z = (angle, [dx, dy], content...)."""

    def __init__(self, in_dim: int = None, latent_dim: int = 2, nb_classes: int = 0, translation: bool = True,
                 seed: int = 0, **kwargs) -> None:
        coord = 3 if translation else 1
        super().__init__(in_dim, latent_dim, nb_classes, coord, **kwargs)
        set_train_rng(seed)
        self.translation = translation
        self.dx_prior = None
        self.phi_prior = None
        self.kdict_ = dc(kwargs)
        self.kdict_["num_iter"] = 0
        self.loss = "mse"

    def elbo_fn(self, x, x_reconstr, *args, **kwargs) -> torch.Tensor:
        return rvae_loss(self.loss, self.in_dim, x, x_reconstr, *args, **kwargs)

    def _default_reparameterize(self) -> bool:
        """True unless ``reparameterize`` was overridden (subclass or instance attribute, as tests do to inject eps)."""
        from ...trainers import viBaseTrainer
        return ("reparameterize" not in self.__dict__
                and getattr(type(self).reparameterize, "__func__", None) is viBaseTrainer.reparameterize.__func__)

    def forward_compute_elbo(self, x: torch.Tensor, y: Optional[torch.Tensor] = None,
                             mode: str = "train") -> torch.Tensor:
        """Same dataflow as rvae.py:110-147: encoder -> reparameterise -> split (phi, dx, z) [-> append the one-hot
        class] -> rotate/translate the coordinate grid -> spatial decoder -> ELBO."""
        with torch.set_grad_enabled(mode != "eval"):
            z_mean, z_logsd = self.encoder_net(x)
            if mode != "eval":
                self.kdict_["num_iter"] += 1
            if y is None and self._default_reparameterize() and z_mean.dtype == torch.float32:
                # the same draw as reparameterize() (one normal_() on the compute device), then ONE kernel for
                # z = mean + sd * eps, the (phi, dx, dy) / content split and the translation prior
                eps = z_mean.new(z_mean.size(0), z_mean.size(1)).normal_()
                theta, z = _LatentFn.apply(z_mean, z_logsd, eps, bool(self.translation), float(self.dx_prior or 0.0))
                x_reconstr = self.decoder_net.forward_grid(self.x_coord, theta, z)
                return self.elbo_fn(x, x_reconstr, z_mean, z_logsd, **self.kdict_)
            z = self.reparameterize(z_mean, torch.exp(z_logsd))
            phi = z[:, :1]
            if self.translation:
                theta = torch.cat((phi, z[:, 1:3] * self.dx_prior), 1)
                z = z[:, 3:]
            else:
                theta = torch.cat((phi, torch.zeros_like(z[:, :2])), 1)
                z = z[:, 1:]
            if y is not None:
                z = torch.cat((z, to_onehot(y, self.nb_classes)), -1)
            # transform_coordinates(x_coord, phi, dx) is applied per pixel inside the decoder kernels
            x_reconstr = self.decoder_net.forward_grid(self.x_coord, theta, z)
            return self.elbo_fn(x, x_reconstr, z_mean, z_logsd, **self.kdict_)

    def fit(self, X_train, y_train=None, X_test=None, y_test=None, loss: str = "mse", **kwargs) -> None:
        self._check_inputs(X_train, y_train, X_test, y_test)
        self.dx_prior = kwargs.get("translation_prior", 0.1)
        self.kdict_["phi_prior"] = kwargs.get("rotation_prior", 0.1)
        for k, v in kwargs.items():
            if k in ["capacity"]:
                self.kdict_[k] = v
        self.compile_trainer((X_train, y_train), (X_test, y_test), **kwargs)
        self.loss = loss
        if self.loss == "ce":                                 # decode() then applies a sigmoid ("prediction" stage)
            self.sigmoid_out = True
            self.metadict["sigmoid_out"] = True
        if kwargs.get("recording", False):
            raise NotImplementedError("manifold recording (matplotlib/torchvision tooling) is out of scope")
        self._fit_loop()
