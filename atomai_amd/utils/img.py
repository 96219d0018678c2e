"""Image helpers on the predict path and between Segmentor predictions and VAE training stacks
(reference: atomai/utils/img.py:112-180, 298-350, 502-551)."""
from typing import Dict, Tuple, Union

import numpy as np


def img_resize(image_data: np.ndarray, rs: Tuple[int], round_: bool = False) -> np.ndarray:
    """Resizes an (n, h, w) stack to ``rs`` (reference: atomai/utils/img.py:20-68, ``img_resize`` + ``cv_resize``) on the
    device: ``amx_aug_resample`` carries OpenCV's INTER_CUBIC and enlarging INTER_AREA arithmetic (restated, UNPINNED
    against cv2 itself — absent from this image; oracle/aug_oracle.py:img_resize is the checker).  Reference quirks kept:
    ``rs`` is swapped when its entries differ (img.py:36-37); a stack that already has the target shape is copied; per
    image INTER_AREA is chosen when the height is below the target's second entry, INTER_CUBIC otherwise (img.py:63-64);
    float64 result.  True area averaging (INTER_AREA with both axes shrunk: non-square targets only) is not provided.
    Precision: the stack is resampled in fp32 on the device whatever its dtype (cv2 keeps CV_64F data in double), so a
    float64 input agrees with a double-precision evaluation of the same arithmetic to ~1e-6 of the data range — a
    TOLERANCE, not bit-level parity (tests/test_oracle_aug_resize.py states it); there is no cv2 in this image to
    generate a golden from."""
    import torch
    from .. import _lib as L
    rs = tuple(int(v) for v in rs)
    if rs[0] != rs[1]:
        rs = (rs[1], rs[0])
    image_data = np.asarray(image_data)
    if image_data.shape[1:3] == rs:
        return image_data.copy()
    n, h, w = image_data.shape
    area = h < rs[1]
    if area and h >= rs[0] and w >= rs[1]:
        raise NotImplementedError("img_resize: cv2.INTER_AREA with both axes shrunk (true area averaging) is not on the "
                                  "device path; it is only reached for non-square targets")
    if torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
    elif L.is_test_backend():
        dev = torch.device("cpu")
    else:
        raise L.AmxError("img_resize runs on the MI355X (amx_aug_resample); there is no CPU fallback")
    out = np.empty((n, rs[0], rs[1]), dtype=np.float64)
    per = max(1, (256 << 20) // (4 * max(h * w, rs[0] * rs[1])))            # frames per upload (<= 256 MB each way)
    for i0 in range(0, n, per):
        x = torch.from_numpy(np.ascontiguousarray(image_data[i0:i0 + per], dtype=np.float32)).to(dev)
        m = x.shape[0]
        y = torch.empty((m, rs[0], rs[1]), dtype=torch.float32, device=dev)
        win = torch.tensor([[0, 0, h, w]] * m, dtype=torch.int32, device=dev)
        L.call("amx_aug_resample", L.ptr(x), L.ptr(y), L.ptr(win), m, h, w, rs[0], rs[1], 2 if area else 1, 0,
               int(bool(round_)), L.stream_ptr(x))
        out[i0:i0 + m] = y.cpu().numpy()
    return out


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


def get_imgstack(imgdata: np.ndarray, coord: np.ndarray, r: int) -> Tuple[np.ndarray]:
    """Square windows of side ``r`` around the given (row, col) centres of ONE image (h, w, c); centres whose window
    sticks out of the image or contains NaN are dropped (img.py:138-180).  Vectorised: one fancy-indexing gather
    instead of a Python loop over the centres."""
    coord = np.asarray(coord)
    if len(coord) == 0:
        return None, None
    r = int(r)
    centres = np.around(coord[:, :2]).astype(np.int64)
    lo = centres - r // 2
    hi = lo + r                                      # odd r: [c - r//2, c + r//2]; even r: [c - r//2, c + r//2 - 1]
    h, w = imgdata.shape[:2]
    ok = (lo[:, 0] >= 0) & (lo[:, 1] >= 0) & (hi[:, 0] <= h) & (hi[:, 1] <= w)
    if not ok.any():
        return None, None
    rows = lo[ok, 0][:, None] + np.arange(r)[None, :]
    cols = lo[ok, 1][:, None] + np.arange(r)[None, :]
    stack = imgdata[rows[:, :, None], cols[:, None, :]]
    kept = coord[ok]
    finite = ~np.isnan(stack.reshape(len(stack), -1)).any(axis=1)
    if not finite.any():
        return None, None
    return stack[finite], kept[finite]


def extract_subimages(imgdata: np.ndarray, coordinates: Union[Dict[int, np.ndarray], np.ndarray],
                      window_size: int, coord_class: int = 0) -> Tuple[np.ndarray]:
    """Sub-images centred on the detected objects of one class, for every frame (img.py:298-350): returns
    (stack, centres, frame numbers) — the usual bridge from ``Segmentor.predict`` output to ``rVAE.fit`` input."""
    single = isinstance(coordinates, np.ndarray)
    per_frame = [np.column_stack((coordinates, np.zeros(len(coordinates))))] if single else list(coordinates.values())
    frames = np.asarray(imgdata)
    if frames.ndim == 2:
        frames = frames[None, :, :, None]
    parts = []                                          # (windows, centres, frame index) of every frame that yields any
    for frame_no, (frame, table) in enumerate(zip(frames, per_frame)):
        table = np.asarray(table)
        wanted = table[table[:, 2] == coord_class, :2]
        windows, centres = get_imgstack(frame, wanted, window_size)
        if windows is not None:
            parts.append((windows, centres, np.full(len(centres), frame_no, dtype=int)))
    if not parts:
        return [], [], []                               # the reference's return value when nothing survives
    return tuple(np.concatenate(col, axis=0) for col in zip(*parts))


def crop_borders(imgdata: np.ndarray, thresh: float = 0) -> np.ndarray:
    """Drops the border rows / columns of an (h, w, c) array whose values are all <= thresh (img.py:502-519)."""
    def crop(img):
        mask = img > thresh
        return img[np.ix_(mask.any(1), mask.any(0))]
    return np.array([crop(imgdata[..., i]) for i in range(imgdata.shape[-1])]).transpose(1, 2, 0)


def get_coord_grid(imgdata: np.ndarray, step: int, return_dict: bool = True):
    """Square grid of coordinates for every image of a stack, in the Locator's output format (img.py:522-551)."""
    if np.ndim(imgdata) == 2:
        imgdata = np.expand_dims(imgdata, axis=0)
    ii, jj = np.meshgrid(np.arange(0, imgdata.shape[1], step), np.arange(0, imgdata.shape[2], step), indexing="ij")
    coord = np.stack((ii.ravel(), jj.ravel()), axis=1)
    if return_dict:
        coord = np.concatenate((coord, np.zeros((coord.shape[0], 1))), axis=-1)
        return {i: coord for i in range(imgdata.shape[0])}
    return np.concatenate([coord for _ in range(imgdata.shape[0])], axis=0)
