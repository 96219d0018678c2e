from .losses import BCEWithLogitsLoss, CrossEntropyLoss, select_loss
from .metrics import IoU
from .vi_losses import elbo_terms, infocapacity, rvae_loss, vae_loss

__all__ = ["select_loss", "CrossEntropyLoss", "BCEWithLogitsLoss", "vae_loss", "rvae_loss", "infocapacity",
           "elbo_terms", "IoU"]
