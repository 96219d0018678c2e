"""CPU restatement of the reference's coordinate extraction (TEST INFRASTRUCTURE ONLY — never imported by the
product path).

Follows atomai/predictors/predictor.py:531-639 (Locator.preprocess / run / rem_edge_coord),
atomai/utils/img.py:554-564 (cv_thresh) and atomai/utils/coords.py:21-34 (find_com).  scipy.ndimage — the
library the reference itself calls — does the labelling; cv2 is absent from this image, so the binary
threshold is restated from its documented semantics (THRESH_BINARY: dst = maxval if src > thresh else 0).
Pinned by tests/golden/locator.npz, produced by the reference's own Locator.run with only cv_thresh
replaced by that one-line restatement (oracle/make_golden.py locator).
"""
from typing import Dict

import numpy as np
from scipy import ndimage


def cv_thresh(img: np.ndarray, threshold: float = 0.5) -> np.ndarray:
    return np.where(img > threshold, 1, 0).astype(img.dtype)


def find_com(img: np.ndarray) -> np.ndarray:
    labels, nlabels = ndimage.label(img)                       # default structure: 4-connectivity
    com = np.array(ndimage.center_of_mass(img, labels, np.arange(nlabels) + 1))
    return com.reshape(com.shape[0], 2)


def rem_edge_coord(coords: np.ndarray, h: int, w: int, dist_edge: int) -> np.ndarray:
    drop = [i for i, c in enumerate(coords)
            if c[0] > h - dist_edge or c[0] < dist_edge or c[1] > w - dist_edge or c[1] < dist_edge]
    return np.delete(coords, np.array(drop, dtype=int), axis=0)


def locate(nn_output: np.ndarray, threshold: float = 0.5, dist_edge: int = 5,
           dim_order: str = "channel_last") -> Dict[int, np.ndarray]:
    """{frame: (n, 3) float64 [row, col, class]} exactly as Locator.run builds it (refine=False)."""
    if nn_output.shape[-1] == 1:
        nn_output = np.concatenate((nn_output, 1 - nn_output), axis=3)
    if dim_order == "channel_first":
        nn_output = np.transpose(nn_output, (0, 2, 3, 1))
    elif dim_order != "channel_last":
        raise NotImplementedError
    out = {}
    for i, frame in enumerate(nn_output):
        coords, cat = np.empty((0, 2)), np.empty((0, 1))
        for ch in range(frame.shape[2] - 1):
            c = rem_edge_coord(find_com(cv_thresh(frame[:, :, ch], threshold)), *nn_output.shape[1:3], dist_edge)
            coords = np.append(coords, c, axis=0)
            cat = np.append(cat, np.zeros((c.shape[0], 1)) + ch, axis=0)
        out[i] = np.concatenate((coords, cat), axis=1)
    return out


def synthetic_maps(rs: np.random.RandomState, B: int, H: int, W: int, C: int, n_blobs: int = 12,
                   noise: float = 0.15) -> np.ndarray:
    """Softmax-like class maps with Gaussian blobs of random size (some touching, some at the border,
    salt noise making irregular / single-pixel components) — inputs for the locator tests and bench."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.zeros((B, H, W, C), dtype=np.float32)
    for b in range(B):
        for c in range(max(C - 1, 1)):
            m = np.zeros((H, W), dtype=np.float32)
            for _ in range(n_blobs):
                cy, cx, s = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(0.8, 3.0)
                m = np.maximum(m, np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))
            m = m + noise * rs.rand(H, W).astype(np.float32) * (rs.rand(H, W) > 0.9)
            out[b, :, :, c] = np.clip(m, 0, 1)
        if C > 1:
            out[b, :, :, C - 1] = np.clip(1 - out[b, :, :, :C - 1].sum(-1), 0, 1)
    return out
