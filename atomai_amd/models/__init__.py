from .segmentor import Segmentor

__all__ = ["Segmentor"]
