"""Image helpers on the predict path (reference: atomai/utils/img.py:112-135)."""
import numpy as np


def img_pad(image_data: np.ndarray, pooling: int) -> np.ndarray:
    """Zero-pads (bottom/right) an (n, h, w) stack until h and w are divisible by ``pooling``.
    Same result as the reference's row-by-row np.concatenate loop (float64 output), in one allocation."""
    pooling = int(pooling)
    n, h, w = image_data.shape
    H = -(-h // pooling) * pooling
    W = -(-w // pooling) * pooling
    if H == h and W == w:
        return image_data
    out = np.zeros((n, H, W), dtype=np.result_type(image_data.dtype, np.float64))
    out[:, :h, :w] = image_data
    return out
