"""Locator: class-probability maps -> blob centres on the device
(reference: atomai/predictors/predictor.py:531-639; cv_thresh utils/img.py:554-564; find_com
utils/coords.py:21-34).

Same constructor / ``run`` contract and the same result — ``{frame: (n, 3) float64 [row, col, class]}`` with the
reference's ordering — but threshold, 4-connected labelling, centres of mass and the border filter run as HIP
kernels over whole chunks of frames (``amx_locate_label`` / ``amx_locate_emit``) instead of a per-frame
cv2 + scipy.ndimage loop on the host.
"""
from typing import Dict, Union

import numpy as np
import torch

from .. import _lib as L


def locate_device(prob: torch.Tensor, threshold: float, dist_edge: int) -> Dict[int, np.ndarray]:
    """Centres for a chunk of NHWC probabilities already resident on the device (or, under the test
    backend, on the host).  Returns {local frame index: (n, 3)}."""
    if prob.ndim != 4 or prob.dtype != torch.float32:
        raise ValueError("expected (B, H, W, C) float32 probabilities")
    prob = prob.contiguous()
    B, H, W, C = prob.shape
    nch = max(C - 1, 1)                         # 1-channel output: background = 1 - p is appended by the reference
    lib = L.load()
    nbytes = lib.amx_locate_workspace_bytes(B, H, W, nch)
    if nbytes < 0:
        raise L.AmxError("chunk too large for int32 labels: pass fewer frames per call")
    dev = prob.device
    work = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    sp = L.stream_ptr(prob)
    L.call("amx_locate_label", L.ptr(prob), B, H, W, C, nch, float(threshold), int(dist_edge), L.ptr(work),
           L.ptr(count), sp)
    n = int(count.item())                       # one 4-byte read-back per chunk sizes the ragged output
    coords = torch.empty((n, 2), dtype=torch.float64, device=dev)
    meta = torch.empty((n, 2), dtype=torch.int32, device=dev)
    if n:
        L.call("amx_locate_emit", L.ptr(work), B, H, W, nch, int(dist_edge), L.ptr(coords), L.ptr(meta), n, sp)
    coords, meta = coords.cpu().numpy(), meta.cpu().numpy()
    table = np.concatenate((coords, meta[:, 1:2].astype(np.float64)), axis=1)
    bounds = np.searchsorted(meta[:, 0], np.arange(B + 1))      # rows are sorted by frame
    return {i: table[bounds[i]:bounds[i + 1]] for i in range(B)}


class Locator:
    """``Locator(threshold=0.5, dist_edge=5, dim_order='channel_last', **kwargs).run(nn_output)``."""

    def __init__(self, threshold: float = 0.5, dist_edge: int = 5, dim_order: str = "channel_last",
                 **kwargs: Union[bool, float]) -> None:
        self.dim_order = dim_order
        self.threshold = threshold
        self.dist_edge = dist_edge
        self.refine = kwargs.get("refine")
        self.d = kwargs.get("d")
        self.device = kwargs.get("device", "cuda" if torch.cuda.is_available() else "cpu")
        self.chunk_bytes = int(kwargs.get("chunk_bytes", 1 << 30))

    def preprocess(self, nn_output: np.ndarray) -> np.ndarray:
        """Channel-last view of the network output.  (The reference also appends a background channel to
        1-channel data, only to skip it again in ``run``; the kernels take the channel count instead.)"""
        if self.dim_order == "channel_first":
            nn_output = np.transpose(nn_output, (0, 2, 3, 1))
        elif self.dim_order != "channel_last":
            raise NotImplementedError('For dim_order, use "channel_first"', 'or "channel_last" (e.g. tensorflow)')
        return nn_output

    def run(self, nn_output: np.ndarray, *args: np.ndarray) -> Dict[int, np.ndarray]:
        if self.refine:
            raise NotImplementedError("peak refinement (per-atom scipy.optimize Gaussian fits) is outside the "
                                      "MI355X hot path of this build")
        nn_output = self.preprocess(np.asarray(nn_output))
        if nn_output.ndim != 4:
            raise ValueError("expected a 4D network output")
        n = len(nn_output)
        per_frame = int(np.prod(nn_output.shape[1:])) * 4 * 8   # probabilities + labelling workspace
        chunk = max(1, min(n, self.chunk_bytes // max(per_frame, 1)))
        out = {}
        for s in range(0, n, chunk):
            x = torch.from_numpy(np.ascontiguousarray(nn_output[s:s + chunk], dtype=np.float32)).to(self.device)
            for i, v in locate_device(x, self.threshold, self.dist_edge).items():
                out[s + i] = v
        return out

    def rem_edge_coord(self, coordinates: np.ndarray, h: int, w: int) -> np.ndarray:
        """Host version of the border filter (predictor.py:621-639), kept for API compatibility."""
        c = np.asarray(coordinates)
        drop = (c[:, 0] > h - self.dist_edge) | (c[:, 0] < self.dist_edge) | \
               (c[:, 1] > w - self.dist_edge) | (c[:, 1] < self.dist_edge)
        return c[~drop]
