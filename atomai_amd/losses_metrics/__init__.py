from .losses import BCEWithLogitsLoss, CrossEntropyLoss, select_loss

__all__ = ["select_loss", "CrossEntropyLoss", "BCEWithLogitsLoss"]
