from .locator import Locator
from .predictor import BasePredictor, SegPredictor

__all__ = ["BasePredictor", "SegPredictor", "Locator"]
