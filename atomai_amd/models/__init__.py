from .dgm import VAE, BaseVAE, rVAE
from .dklgp import dklGPR
from .segmentor import Segmentor
from .loaders import (load_ensemble, load_model, load_pretrained_model, load_seg_model,  # noqa: E402
                      load_vae_model)

__all__ = ["Segmentor", "BaseVAE", "VAE", "rVAE", "dklGPR", "load_model", "load_ensemble", "load_pretrained_model",
           "load_seg_model", "load_vae_model"]
