"""On-the-fly augmentation of the device-resident mini-batch (reference: atomai/transforms/imaug.py:20-432).

The reference runs numpy / skimage / scipy / cv2 in float64 on the CPU for every mini-batch and re-uploads the result
(hook: atomai/trainers/trainer.py:339-341); at > 1 k images/s that would starve the GPU.  Here the batch stays in HBM
and the steps are HIP kernels (csrc/aug.hip).  Same step order and parameter ranges as ``datatransform.run``:

    (x - min) / ptp -> rotation -> gauss_noise -> jitter -> poisson_noise -> salt_and_pepper -> blur -> contrast
    -> background
    -> [drop image-label pairs that lost a class] -> (x - min) / ptp

* The per-image scalar parameters (flip type, noise level, gamma, background centre / widths / amplitude) are drawn on
  the host from ``np.random.RandomState(seed)`` in the reference's order — with the same seed they are the reference's
  values, except after ``poisson_noise``, whose per-pixel ``np.random.poisson`` draws consume the reference's global
  stream by an amount that cannot be reproduced.
* Per-pixel randomness comes from a counter-based Philox generator in the kernel (skimage draws its own from an unseeded
  ``default_rng``, so the reference is not reproducible there either); ``fields=`` injects explicit noise fields so that
  the arithmetic can be compared with the reference element by element (tests/golden/augment.npz).
* ``jitter`` (rows rolled by Poisson-distributed shifts, imaug.py:123-135): the N x H shifts are drawn on the host from
  the same stream (``scipy.stats.poisson.rvs`` draws from numpy's global state, i.e. they are the reference's shifts
  for the same seed) and applied as an index map inside the point pass.
* ``zoom`` / ``resize`` (imaug.py:195-227, 276-300): the window sizes / output size are drawn on the host in the
  reference's order (same seed -> the reference's choices) and ``cv2.resize`` runs as a resampling kernel
  (``amx_aug_resample``): INTER_CUBIC (a = -0.75) for zoom; for resize INTER_LINEAR, because the reference's
  ``cv2.resize(img, (w, h), rs_method)`` passes its method in the position of ``dst`` and so never changes the default
  interpolation.  OpenCV's arithmetic is restated from its documentation — UNPINNED against cv2 itself (absent in this
  image); everything around it (draw order, crop windows, size formulas, rounding, the squeeze / drop rule) is pinned to
  the reference's own code run over that restatement (tests/golden/augment_geom.npz).  Class maps travel as the K
  one-hot planes the reference resamples and are squeezed back with its ``sum_c c * mask_c`` rule.
* ``custom_transform`` (imaug.py:79, 323-324: a user callable applied first, to the normalised images and the one-hot
  masks) is arbitrary HOST code: the batch is normalised on the device, handed to the callable as the numpy arrays the
  reference passes — images (N, H, W) float64, masks (N, H, W, C) float64 — and the result is uploaded again; everything
  after it runs on the device as usual (no second normalisation before the final one, as in the reference).  The masks it
  returns must still be one-hot / binary; the batch size and image size may change.
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from .. import _lib as L

_NP = 12
_UNSUPPORTED = ()


def _minmax(x: torch.Tensor) -> torch.Tensor:
    n = x.numel()
    work = torch.empty(2 * L.load().amx_aug_minmax_blocks(n), dtype=torch.float32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    L.call("amx_aug_minmax", L.ptr(x), n, L.ptr(work), L.ptr(out), L.stream_ptr(x))
    return out


def _unique_counts(x: torch.Tensor) -> np.ndarray:
    """len(np.unique(image)) per image (imaug.py:145) — a library sort, host read-back of N integers."""
    s, _ = torch.sort(x.reshape(x.shape[0], -1), dim=1)
    return (1 + (s[:, 1:] != s[:, :-1]).sum(1)).cpu().numpy()


class datatransform:
    """Device-side counterpart of the reference's ``datatransform`` for (N, H, W) image batches and integer / binary
    label maps.  ``run(images, labels, fields=None)`` returns the augmented (images (N', 1, H, W), labels)."""

    def __init__(self, n_channels: int = None, seed: Optional[int] = None, **kwargs) -> None:
        bad = [k for k in _UNSUPPORTED if kwargs.get(k)]
        if bad:
            raise NotImplementedError(f"augmentation {bad} is not available on the device path (host-code "
                                      "transforms, see atomai_amd/transforms/imaug.py)")
        self.ch = n_channels
        if n_channels is not None and n_channels > 31:
            # the label kernels keep one presence bit per class in a 32-bit word (csrc/aug.hip: amx_aug_labels / _squeeze)
            raise NotImplementedError(f"datatransform on the device supports up to 31 classes (n_channels={n_channels})")
        self.custom_transform = kwargs.get("custom_transform")
        if self.custom_transform is not None and not callable(self.custom_transform):
            raise TypeError("custom_transform must be a callable (images, targets) -> (images, targets)")
        rng = lambda key, dflt: (dflt if kwargs.get(key) is True else kwargs.get(key))   # noqa: E731
        self.rotation = kwargs.get("rotation")
        self.background = kwargs.get("background")
        self.gauss = rng("gauss_noise", [0, 50])
        self.jitter = rng("jitter", [0, 50])
        self.poisson = rng("poisson_noise", [30, 40])
        self.salt_and_pepper = rng("salt_and_pepper", [0, 50])
        self.blur = rng("blur", [1, 50])
        self.contrast = rng("contrast", [5, 20])
        self.zoom = 2 if kwargs.get("zoom") is True else kwargs.get("zoom")
        self.resize = [2, 1.5] if kwargs.get("resize") is True else kwargs.get("resize")
        self.rs = np.random.RandomState(seed)                 # np.random.seed(seed) of imaug.py:106
        self.seed = 0 if seed is None else int(seed)
        self.params = None

    @staticmethod
    def _is_range(v) -> bool:
        return isinstance(v, (list, tuple))

    # ------------------------------------------------------------------ host: scalar draws in the reference's order
    def draw_geometry(self, n: int, h: int, w: int):
        """rotation codes, zoom windows, resize target — the first draws of ``run`` (imaug.py:319-325) — and the image
        size the remaining steps will see."""
        rs = self.rs
        flips = np.full(n, 4.0)
        if self.rotation:
            for i in range(n):
                ft = rs.randint(-1, 3)                       # 3 is never drawn (imaug.py:267); 2 = rot90 ccw if square
                flips[i] = ft if (ft != 2 or h == w) else 1  # cv2.flip(img, 2) on a non-square image flips horizontally
        zv = None
        if self.zoom:                                        # imaug.py:202-208: one np.random.choice per image
            S = min(h, w)
            cand = np.arange(int(S // self.zoom), S + 8, 8)
            cand = cand[cand <= S]
            zv = np.array([rs.choice(cand) for _ in range(n)], dtype=np.int64)
            h = w = S
        out_hw = None
        if self._is_range(self.resize):                      # imaug.py:283-292: ONE size for the whole batch
            d, u = 1 / self.resize[0], self.resize[1]
            s_, p_ = 0.03, 8
            while np.round((h * s_), 7) % p_ != 0 and np.round((w * s_), 7) % p_ != 0:
                s_ += 1e-5
            rs_h = (np.arange(d, u, s_) * h).astype(np.int64)
            rs_w = (np.arange(d, u, s_) * w).astype(np.int64)
            k = rs.randint(len(rs_h))
            if (h, w) != (rs_h[k], rs_w[k]):
                out_hw = (int(rs_h[k]), int(rs_w[k]))
                h, w = out_hw
        return flips, zv, out_hw, h, w

    def draw(self, n: int, h: int, w: int, flips=None):
        rs = self.rs
        P = np.zeros((n, _NP), dtype=np.float64)        # float64 here (the oracle's input), fp32 on the device
        P[:, 0] = 4
        extra = {}
        if flips is not None:
            P[:, 0] = flips
        elif self.rotation:
            for i in range(n):
                ft = rs.randint(-1, 3)                       # 3 is never drawn (imaug.py:267); 2 = rot90 ccw if square
                P[i, 0] = ft if (ft != 2 or h == w) else 1   # cv2.flip(img, 2) on a non-square image flips horizontally
        if self._is_range(self.gauss):
            for i in range(n):
                P[i, 1] = np.sqrt(1e-4 * rs.randint(self.gauss[0], self.gauss[1]))
        if self._is_range(self.jitter):                       # one level per image, then H poisson shifts (imaug.py:132-133)
            extra["jitter"] = np.stack([rs.poisson(rs.randint(self.jitter[0], self.jitter[1]) / 10, size=h)
                                        for _ in range(n)]).astype(np.int32)
        if self._is_range(self.poisson):
            extra["poisson_l"] = np.array([rs.randint(self.poisson[0], self.poisson[1]) for _ in range(n)])
        if self._is_range(self.salt_and_pepper):
            for i in range(n):
                P[i, 3] = rs.randint(self.salt_and_pepper[0], self.salt_and_pepper[1]) * 1e-3
        if self._is_range(self.blur):
            extra["blur_sigma"] = np.array([rs.randint(self.blur[0], self.blur[1]) * 5e-2 for _ in range(n)])
        if self._is_range(self.contrast):
            for i in range(n):
                P[i, 4] = rs.randint(self.contrast[0], self.contrast[1]) / 10
        if self.background:
            for i in range(n):
                x0 = rs.randint(0, h - h // 4)
                y0 = rs.randint(0, w - w // 4)
                a, b = rs.randint(10, 20, 2) / 10
                fwhm = rs.randint(min(h, w) // 4, min(h, w) - min(h, w) // 2)
                amp = 0.05 * rs.randint(-10, 10)
                P[i, 5:11] = (x0, y0, a, b, fwhm, amp)
        return P, extra

    # ------------------------------------------------------------------ device
    def _point(self, x, P, mnmx=None, fields=None, jitter=None):
        fields = fields or {}
        N, H, W = x.shape
        dev = x.device
        y = torch.empty_like(x)
        Pd = torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32)).to(dev)
        f = lambda k: None if fields.get(k) is None else fields[k].to(dev).float().contiguous()   # noqa: E731
        keep = [f("gauss"), f("poisson"), f("sp_flip"), f("sp_salt")]
        jd = None if jitter is None else torch.from_numpy(np.ascontiguousarray(jitter, dtype=np.int32)).to(dev)
        L.call("amx_aug_point", L.ptr(x), L.ptr(y), L.ptr(Pd), L.ptr(mnmx), L.ptr(keep[0]), L.ptr(keep[1]),
               L.ptr(keep[2]), L.ptr(keep[3]), L.ptr(jd), N, H, W, self.seed, L.stream_ptr(x))
        return y

    @staticmethod
    def _resample(x, win, out_hw, mode: int, clip01: bool, round_out: bool):
        """x (M, Hs, Ws) fp32 planes, win (M, 4) int32 source windows -> (M, Hd, Wd)."""
        M, Hs, Ws = x.shape
        y = torch.empty((M,) + tuple(out_hw), dtype=torch.float32, device=x.device)
        wd = torch.from_numpy(np.ascontiguousarray(win, dtype=np.int32)).to(x.device)
        L.call("amx_aug_resample", L.ptr(x), L.ptr(y), L.ptr(wd), M, Hs, Ws, int(out_hw[0]), int(out_hw[1]), mode,
               int(clip01), int(round_out), L.stream_ptr(x))
        return y

    def _host_transform(self, x: torch.Tensor, targets: torch.Tensor):
        """``custom_transform`` (imaug.py:323-324): (x - min) / ptp on the device, then the user's callable on the host
        with the arrays the reference hands it — images (N, H, W) float64, masks (N, H, W, C) float64 (one-hot planes of a
        class map, the mask itself for one class) — and the result back on the device as (images fp32, class map int64 /
        binary mask fp32 in the layout `targets` came in)."""
        N, H, W = x.shape
        dev = x.device
        P = np.zeros((N, _NP))
        P[:, 0] = 4                                           # no flip: normalisation only
        xi = self._point(x, P, _minmax(x), None, None).cpu().numpy().astype(np.float64)
        multi = bool(self.ch and self.ch > 1)
        tnp = targets.detach().cpu().numpy()
        if multi:
            t = np.eye(self.ch)[(tnp[:, 0] if tnp.ndim == 4 else tnp).astype(np.int64)]          # (N, H, W, C)
        else:
            t = (tnp[:, 0] if tnp.ndim == 4 else tnp).astype(np.float64)[..., None]
        xi, t = self.custom_transform(xi, t)
        xi, t = np.asarray(xi), np.asarray(t)
        if xi.ndim != 3 or t.ndim != 4 or t.shape[:3] != xi.shape or t.shape[-1] != (self.ch if multi else 1):
            raise ValueError(f"custom_transform must return images (N, H, W) and masks (N, H, W, {self.ch if multi else 1}); "
                             f"got {xi.shape} and {t.shape}")
        if multi:
            # The reference hands whatever the callable returned to its geometric steps (each followed by np.around on
            # the masks) and finally to squeeze_channels: label = sum_c c * mask_c, the pair is kept iff exactly C
            # distinct labels occur (imaug.py:361-393) — it never raises.  Same here: masks are rounded, squeezed, and a
            # frame whose labels leave [0, C-1] (overlapping channels) is dropped, as the reference's rule drops a frame
            # that shows all C classes plus such a value.  (A frame that LACKS a class and has an out-of-range label in
            # its place survives in the reference with that bogus label; here it is dropped too — the device path
            # carries class maps, which cannot hold it.)  Uncovered pixels become class 0, as in the reference.
            cls = np.tensordot(np.around(t), np.arange(self.ch, dtype=np.float64), axes=([3], [0]))   # sum_c c * mask_c
            flat = cls.reshape(len(cls), -1)
            ok = (flat.min(1) >= 0) & (flat.max(1) <= self.ch - 1)
            if not ok.all():
                xi, cls = xi[ok], cls[ok]
            if not len(xi):
                raise RuntimeError("custom_transform left no usable frame: every returned mask has pixels whose channels "
                                   "overlap (label sum_c c * mask_c outside [0, n_channels - 1]); the reference's "
                                   "squeeze_channels rule drops such frames (transforms/imaug.py:361-393)")
            x = torch.from_numpy(np.ascontiguousarray(xi, dtype=np.float32)).to(dev)
            tt = torch.from_numpy(cls.astype(np.int64)).to(dev)
            return x, (tt[:, None] if targets.ndim == 4 else tt)
        x = torch.from_numpy(np.ascontiguousarray(xi, dtype=np.float32)).to(dev)
        tt = torch.from_numpy(np.ascontiguousarray(t[..., 0], dtype=np.float32)).to(dev)
        return x, (tt[:, None] if targets.ndim == 4 else tt)

    def _geometry(self, planes, zv, out_hw, reps: int, is_mask: bool):
        """zoom (centred zv x zv crop -> short side, INTER_CUBIC) then resize (whole frame -> out_hw, INTER_LINEAR as the
        reference's call executes) of (N * reps, H, W) planes; images are clipped after the zoom, masks rounded."""
        M, H, W = planes.shape
        if zv is not None:
            S = min(H, W)
            z = np.repeat(zv, reps)
            win = np.stack([H // 2 - z // 2, W // 2 - z // 2, 2 * (z // 2), 2 * (z // 2)], 1)
            planes = self._resample(planes, win, (S, S), 1, not is_mask, is_mask)
            H = W = S
        if out_hw is not None:
            win = np.tile(np.array([[0, 0, H, W]]), (M, 1))
            planes = self._resample(planes, win, out_hw, 0, False, is_mask)
        return planes

    def run(self, images: torch.Tensor, targets: torch.Tensor, fields: dict = None) -> Tuple[torch.Tensor]:
        """images (N, H, W) or (N, 1, H, W) fp32 on the device; targets (N, H, W) int64 class maps or (N, 1, H, W)
        fp32 binary masks.  ``fields`` (tests): {'gauss','poisson','sp_flip','sp_salt'} -> (N, H, W) tensors, 'jitter' -> (N, H) ints."""
        x = images[:, 0] if images.ndim == 4 else images
        x = x.float().contiguous()
        normalised = False
        if self.custom_transform is not None:
            x, targets = self._host_transform(x, targets)
            normalised = True                                 # (the reference normalises ONCE before the callable)
        N, H, W = x.shape
        H0, W0 = H, W
        flips, zv, out_hw, H, W = self.draw_geometry(N, H, W)
        geo = zv is not None or out_hw is not None
        self.geometry = {"flips": flips, "zoom_zv": zv, "resize_hw": out_hw}
        if geo:
            # (x - min) / ptp and the flips first, then the resampling steps; the point passes below then run on the
            # resampled batch with the normalisation and the flip switched off
            P0 = np.zeros((N, _NP))
            P0[:, 0] = flips
            x = self._point(x, P0, None if normalised else _minmax(x), None, None)
            x = self._geometry(x, zv, out_hw, 1, False)
            P, extra = self.draw(N, H, W, flips=np.full(N, 4.0))
        else:
            P, extra = self.draw(N, H, W, flips=flips)
        self.params, self.extra = P, extra
        fields = dict(fields or {})
        zero = np.zeros_like(P)
        zero[:, 0] = 4
        # ---- pass A: normalise, rotate, gaussian noise
        PA = zero.copy()
        PA[:, 0:2] = P[:, 0:2]
        rest = P.copy()
        rest[:, 0] = 4
        rest[:, 1] = 0
        # (jitter moves pixels BETWEEN the gaussian step and the later per-pixel steps: those run in their own pass)
        need_split = "poisson_l" in extra or "blur_sigma" in extra or "jitter" in extra
        if not need_split:
            PA = P                                            # everything in ONE pass
        if "jitter" in fields:                                # tests: explicit shifts
            extra["jitter"] = np.asarray(fields["jitter"], dtype=np.int32)
        x = self._point(x, PA, None if (geo or normalised) else _minmax(x), fields, extra.get("jitter"))
        if need_split:
            # ---- pass B: poisson (its scale needs the number of distinct values of the image so far), salt & pepper
            PB = zero.copy()
            PB[:, 3] = rest[:, 3]
            if "poisson_l" in extra:
                vals = _unique_counts(x)
                PB[:, 2] = (50.0 / extra["poisson_l"]) ** np.ceil(np.log2(vals))
                extra["poisson_vals"] = PB[:, 2].copy()
            if PB[:, 2:4].any():
                x = self._point(x, PB, None, fields)
            if "blur_sigma" in extra:
                sg = torch.from_numpy(extra["blur_sigma"].astype(np.float32)).to(x.device)
                t = torch.empty_like(x)
                L.call("amx_aug_blur", L.ptr(x), L.ptr(t), L.ptr(sg), N, H, W, 0, L.stream_ptr(x))
                L.call("amx_aug_blur", L.ptr(t), L.ptr(x), L.ptr(sg), N, H, W, 1, L.stream_ptr(x))
            # ---- pass C: contrast, background
            PC = zero.copy()
            PC[:, 4:] = rest[:, 4:]
            if PC[:, 4].any() or PC[:, 10].any():
                x = self._point(x, PC, None, None)
        # ---- labels: same flips / rotations; drop pairs in which a class disappeared (squeeze_channels)
        keep = None
        if targets.dtype == torch.int64:
            t = targets.contiguous()
            out_t = torch.empty_like(t)
            present = torch.zeros(N, dtype=torch.int32, device=t.device)
            PL = np.zeros((N, _NP), dtype=np.float32)
            PL[:, 0] = flips
            Pd = torch.from_numpy(PL).to(t.device)
            L.call("amx_aug_labels", L.ptr(t), L.ptr(out_t), L.ptr(Pd), L.ptr(present), N, H0, W0, L.stream_ptr(t))
            targets = out_t
            if geo:
                if not self.ch or self.ch < 2:
                    raise NotImplementedError("zoom / resize of integer class maps needs n_channels >= 2")
                K = self.ch
                masks = torch.empty((N * K, H0, W0), dtype=torch.float32, device=t.device)
                L.call("amx_aug_onehot", L.ptr(out_t), L.ptr(masks), N, K, H0 * W0, L.stream_ptr(t))
                masks = self._geometry(masks, zv, out_hw, K, True)
                targets = torch.empty((N, H, W), dtype=torch.int64, device=t.device)
                values = torch.zeros(N, dtype=torch.int32, device=t.device)
                L.call("amx_aug_squeeze", L.ptr(masks), L.ptr(targets), L.ptr(values), N, K, H * W, L.stream_ptr(t))
                # squeeze_channels (imaug.py:390): the pair survives iff exactly K distinct label values occur
                keep = np.array([bin(int(v) & 0xffffffff).count("1") == K for v in values.cpu().numpy()])
            elif self.ch and self.ch > 1:
                full = (1 << self.ch) - 1
                keep = (present.cpu().numpy() & full) == full         # host sync: the batch size may change
        else:
            t = (targets[:, 0] if targets.ndim == 4 else targets).float().contiguous()
            PF = zero.copy()
            PF[:, 0] = flips
            t = self._point(t, PF, None, None)
            if geo:
                t = self._geometry(t, zv, out_hw, 1, True)
            targets = t[:, None] if targets.ndim == 4 else t
        if keep is not None and not keep.all():
            idx = torch.from_numpy(np.nonzero(keep)[0]).to(x.device)
            x, targets = x.index_select(0, idx).contiguous(), targets.index_select(0, idx).contiguous()
        if not x.shape[0]:
            # the reference's squeeze_channels drops a pair as soon as one class is absent after the transform; it then
            # fails inside the next conv with a shape error — say what happened instead
            raise RuntimeError("data augmentation dropped every image of the batch: after the geometric transform at least "
                               "one label class is missing from each frame (the reference's squeeze_channels rule, "
                               "transforms/imaug.py:361-393); use larger frames / zoom windows or fewer classes")
        mm = _minmax(x)                                       # kept referenced until the launch is enqueued
        L.call("amx_aug_renorm", L.ptr(x), x.numel(), L.ptr(mm), L.stream_ptr(x))
        return x[:, None], targets


def seg_augmentor(nb_classes: int, **kwargs) -> Optional[Callable]:
    """``augmentor(images, labels, seed)`` for BaseTrainer.data_augmentation / Segmentor.fit(..., rotation=True, ...)
    (imaug.py:398-432); None when no augmentation keyword is given."""
    auglist = ["custom_transform", "zoom", "gauss_noise", "jitter", "poisson_noise", "contrast", "salt_and_pepper",
               "blur", "resize", "rotation", "background"]
    augdict = {k: kwargs[k] for k in auglist if k in kwargs}
    if len(augdict) == 0:
        return None
    datatransform(nb_classes, 0, **augdict)                  # unsupported keys raise now, not at the first batch

    def augmentor(images, labels, seed):
        if not images.is_cuda and torch.cuda.is_available():      # batches kept on the host (memory_alloc exceeded)
            images = images.cuda()
        dev = images.device
        if not (images.is_cuda or L.is_test_backend()):
            raise L.AmxError("on-device augmentation needs the batch on the MI355X")
        dt = datatransform(nb_classes, seed, **augdict)
        return dt.run(images, labels.to(dev))

    return augmentor
