from .dklgpr import dklGPR

__all__ = ["dklGPR"]
