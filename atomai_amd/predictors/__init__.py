from .epredictor import EnsemblePredictor, ensemble_locate
from .locator import Locator
from .predictor import BasePredictor, SegPredictor

__all__ = ["BasePredictor", "SegPredictor", "Locator", "EnsemblePredictor", "ensemble_locate"]
