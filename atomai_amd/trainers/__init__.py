from .trainer import BaseTrainer, SegTrainer

__all__ = ["BaseTrainer", "SegTrainer"]
