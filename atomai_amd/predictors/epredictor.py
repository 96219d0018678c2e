"""Prediction with an ensemble of trained weights (reference: atomai/predictors/epredictor.py:21-330).

``EnsemblePredictor(skeleton, ensemble, nb_classes=...).predict(data)`` -> (mean, variance) over the members, and
``ensemble_locate`` -> per-frame mean / variance of every detected coordinate.  Image-to-image (segmentation) models
only; every member's forward runs on the HIP engine (probabilities from the fused softmax / sigmoid head).
"""
from typing import Dict, Tuple, Type, Union

import numpy as np
import torch

from ..nets.fcnn import _HipNet, predict_proba
from ..utils import get_downsample_factor, torch_format_image
from .locator import Locator
from .predictor import BasePredictor


class EnsemblePredictor(BasePredictor):
    def __init__(self, skeleton: Type[torch.nn.Module], ensemble: Dict[int, Dict[str, torch.Tensor]],
                 data_type: str = "image", output_type: str = "image", nb_classes: int = None,
                 in_dim: Tuple[int] = None, out_dim: Tuple[int] = None, **kwargs: Union[str, Tuple[int]]) -> None:
        super().__init__()
        if output_type not in ["image", "spectra"]:
            raise TypeError("Supported output types are 'image' and 'spectra'")
        if data_type != "image" or output_type != "image":
            raise NotImplementedError("spectra (ImSpec) ensembles are outside the MI355X hot path of this build")
        self.device = "cpu"
        if kwargs.get("use_gpu", True) and torch.cuda.is_available():
            self.device = kwargs.get("device") or "cuda"
        self.model = skeleton
        self.ensemble = ensemble
        self.data_type, self.output_type = data_type, output_type
        self.nb_classes = nb_classes
        self.in_dim, self.out_dim = in_dim, out_dim
        self.downsample_factor = None
        self.logits = kwargs.get("logits", True)
        self.output_shape = kwargs.get("output_shape")
        verbose = kwargs.get("verbose", 1)
        self.everbose = bool(verbose)
        self.verbose = verbose > 1

    def _set_output_shape(self, data) -> None:
        ch = self.nb_classes if self.nb_classes else 1
        self.output_shape = (len(data), ch, *data.shape[2:])

    def preprocess(self, data: np.ndarray, norm: bool = True) -> torch.Tensor:
        if data.ndim == 2:
            data = data[np.newaxis, ...]
        return torch_format_image(data, norm)

    def _member_probabilities(self, data: torch.Tensor) -> torch.Tensor:
        """Channel-first probabilities of the currently loaded member."""
        nclasses = self.nb_classes or 0
        if isinstance(self.model, _HipNet) and self.logits and nclasses >= 1:
            self.model.eval()
            return predict_proba(self.model, data.to(self.device)).permute(0, 3, 1, 2)
        prob = self.forward_(data)
        if self.logits:
            if nclasses > 1:
                prob = torch.softmax(prob, dim=1)
            elif self.nb_classes == 1:
                prob = torch.sigmoid(prob)
        elif nclasses > 1:
            prob = torch.exp(prob)
        return prob

    def ensemble_forward(self, data: torch.Tensor, out_shape: Tuple[int], num_batches: int = 1) -> np.ndarray:
        """ALL predictions (n_models x n_samples x ...), float64 as in the reference."""
        epred = np.zeros((len(self.ensemble), *out_shape))
        for i, m in enumerate(self.ensemble.values()):
            self.model.load_state_dict(m)
            self._model2device()
            if num_batches > 1:
                bs = max(1, len(data) // num_batches)
                for s in range(0, len(data), bs):
                    epred[i, s:s + bs] = self._member_probabilities(data[s:s + bs]).cpu().numpy()
            else:
                epred[i] = self._member_probabilities(data).cpu().numpy()
        return epred

    def ensemble_forward_(self, data: torch.Tensor, out_shape: Tuple[int]) -> Tuple[np.ndarray]:
        epred = self.ensemble_forward(data, out_shape)
        return np.mean(epred, axis=0), np.var(epred, axis=0)

    def ensemble_batch_predict(self, data, num_batches: int = 10) -> Tuple[np.ndarray]:
        batch_size = len(data) // num_batches
        if batch_size < 1:
            num_batches = batch_size = 1
        mean, var = np.zeros(self.output_shape), np.zeros(self.output_shape)
        i = 0
        for i in range(num_batches):
            if self.everbose:
                print("\rBatch {}/{}".format(i + 1, num_batches), end="")
            sl = slice(i * batch_size, (i + 1) * batch_size)
            mean[sl], var[sl] = self.ensemble_forward_(data[sl], (batch_size, *self.output_shape[1:]))
        rest = data[(i + 1) * batch_size:]
        if len(rest) > 0:
            mean[(i + 1) * batch_size:], var[(i + 1) * batch_size:] = self.ensemble_forward_(
                rest, (len(rest), *self.output_shape[1:]))
        return mean, var

    def predict(self, data: np.ndarray, num_batches: int = 10, format_out: str = "channel_last",
                norm: bool = True) -> Tuple[np.ndarray]:
        if format_out not in ["channel_first", "channel_last"]:
            raise ValueError("Specify channel_last or channel_first output format")
        data = self.preprocess(data, norm)
        if not self.output_shape:
            self._set_output_shape(data)
        if self.downsample_factor is None:
            self._model2device()                 # the mock forward below runs HIP kernels: the net must be on the GPU
            self.downsample_factor = get_downsample_factor(self.model)
        mean, var = self.ensemble_batch_predict(data, num_batches)
        if format_out == "channel_last":
            tr = (0, *(np.arange(mean.ndim - 2) + 2), 1)
            return mean.transpose(tr), var.transpose(tr)
        return mean, var


def cluster_coord(coord_class_dict, eps: float, min_samples: int = 10) -> Tuple[np.ndarray]:
    """DBSCAN clustering of coordinates collapsed over the members (atomai/utils/coords.py:304-347)."""
    from sklearn import cluster
    coordinates_all = np.empty((0, 3))
    for k in range(len(coord_class_dict)):
        coordinates_all = np.append(coordinates_all, coord_class_dict[k], axis=0)
    labels = cluster.DBSCAN(eps=eps, min_samples=min_samples).fit(coordinates_all[:, :2]).labels_
    clusters, clusters_var, clusters_mean = [], [], []
    for lab in np.unique(labels)[1:]:
        coord = coordinates_all[np.where(labels == lab)]
        clusters.append(coord)
        clusters_mean.append(np.mean(coord[:, :2], axis=0))
        clusters_var.append(np.var(coord[:, :2], axis=0))
    return np.array(clusters, dtype=object), np.array(clusters_mean), np.array(clusters_var)


def ensemble_locate(nn_output_ensemble: np.ndarray, **kwargs) -> Tuple[Dict, Dict]:
    """Mean and variance of every detected coordinate over the members' predictions (5D input:
    members x frames x H x W x C); epredictor.py:297-330."""
    eps = kwargs.get("eps", 0.5)
    thresh = kwargs.get("threshold", 0.5)
    loc = Locator(thresh)
    coord_mean_all, coord_var_all = {}, {}
    for i in range(nn_output_ensemble.shape[1]):
        members = loc.run(np.ascontiguousarray(nn_output_ensemble[:, i], dtype=np.float32))   # one launch set
        _, coord_mean, coord_var = cluster_coord(members, eps)
        coord_mean_all[i], coord_var_all[i] = coord_mean, coord_var
    return coord_mean_all, coord_var_all
