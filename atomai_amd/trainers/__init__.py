from .etrainer import BaseEnsembleTrainer, EnsembleTrainer
from .gptrainer import dklGPTrainer
from .trainer import BaseTrainer, SegTrainer
from .vitrainer import viBaseTrainer

__all__ = ["BaseTrainer", "SegTrainer", "viBaseTrainer", "dklGPTrainer", "BaseEnsembleTrainer", "EnsembleTrainer"]
