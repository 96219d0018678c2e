from .dgm import VAE, BaseVAE, rVAE
from .dklgp import dklGPR
from .segmentor import Segmentor

__all__ = ["Segmentor", "BaseVAE", "VAE", "rVAE", "dklGPR"]
