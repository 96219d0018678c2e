from .dgm import VAE, BaseVAE, rVAE
from .segmentor import Segmentor

__all__ = ["Segmentor", "BaseVAE", "VAE", "rVAE"]
