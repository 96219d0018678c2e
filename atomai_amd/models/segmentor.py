"""Segmentor: sklearn-like user API for semantic segmentation (reference: atomai/models/segmentor.py:10-207)."""
from typing import Tuple, Type, Union

import torch

from ..predictors import SegPredictor
from ..trainers import SegTrainer
from ..transforms import seg_augmentor
from ..utils import get_downsample_factor


class Segmentor(SegTrainer):
    """``Segmentor(model='Unet'|'dilnet', nb_classes, **kwargs).fit(...).predict(...)``."""

    def __init__(self, model: Type[Union[str, torch.nn.Module]] = "Unet", nb_classes: int = 1,
                 **kwargs) -> None:
        super().__init__(model, nb_classes, **kwargs)
        self.downsample_factor = None

    def fit(self, X_train, y_train, X_test=None, y_test=None, loss: str = 'ce', optimizer=None,
            training_cycles: int = 1000, batch_size: int = 32, compute_accuracy: bool = False,
            full_epoch: bool = False, swa: bool = False, perturb_weights: bool = False, **kwargs):
        """Compiles the trainer and trains (segmentor.py:61-149).  On-the-fly augmentation keywords (rotation, zoom,
        resize, gauss_noise, jitter, poisson_noise, salt_and_pepper, blur, contrast, background) run as HIP kernels on
        the resident batch (transforms/imaug.py); custom_transform raises.  ``distributed=True``: one process per GPU,
        this rank trains on its shard of ``X_train`` (parallel.py)."""
        self.compile_trainer((X_train, y_train, X_test, y_test), loss, optimizer, training_cycles,
                             batch_size, compute_accuracy, full_epoch, swa, perturb_weights, **kwargs)
        self.augment_fn = seg_augmentor(self.nb_classes, **kwargs)
        _ = self.run()

    def predict(self, imgdata, refine: bool = False, logits: bool = True, resize: Tuple[int, int] = None,
                compute_coords: bool = True, **kwargs):
        """Applies the (trained) model to new data (segmentor.py:151-200)."""
        if self.downsample_factor is None:
            self.downsample_factor = get_downsample_factor(self.net)
        use_gpu = self.device == 'cuda'
        return SegPredictor(self.net, refine, resize, use_gpu, logits, nb_classes=self.nb_classes,
                            downsampling=self.downsample_factor, **kwargs).run(imgdata, compute_coords, **kwargs)

    def load_weights(self, filepath: str) -> None:
        self.net.load_state_dict(torch.load(filepath, map_location=self.device))
