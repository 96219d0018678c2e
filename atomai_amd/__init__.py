"""atomai_amd — MI355X-native hot path of pycroscopy/atomai behind the reference's own Python API.

    import atomai_amd as aoi
    model = aoi.models.Segmentor(nb_classes=3); model.fit(X, y, Xt, yt, training_cycles=1000)

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed = RCCL).  All arithmetic:
hand-written HIP kernels for gfx950 in atomai_amd/csrc, reached through the C ABI of
include/atomai_amd.h.  There is no CPU fallback.
"""
from . import losses_metrics, models, nets, predictors, trainers, transforms, utils  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__version__ = "0.1.0"
