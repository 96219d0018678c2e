"""`not gpu` tier: the numpy restatement of cv2.resize in oracle/aug_oracle.py (cv2 is absent: UNPINNED against cv2
itself) against an independent implementation that documents the same conventions — torch's
F.interpolate(align_corners=False): 'bicubic' (A = -0.75, clamped taps, "matching OpenCV") and 'bilinear'."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import aug_oracle as ao


@pytest.mark.parametrize("mode,tmode", [("cubic", "bicubic"), ("linear", "bilinear")])
@pytest.mark.parametrize("src,dst", [((24, 24), (32, 32)), ((17, 23), (40, 31)), ((64, 64), (45, 45)), ((9, 30), (9, 30)),
                                     ((40, 40), (16, 56))])
def test_cv_resize_restatement_agrees_with_torch_interpolate(mode, tmode, src, dst):
    rs = np.random.RandomState(src[0] * 100 + dst[1])
    img = rs.rand(*src)
    got = ao.cv_resize(img, dst, mode)
    ref = F.interpolate(torch.from_numpy(img)[None, None], size=dst, mode=tmode, align_corners=False)[0, 0].numpy()
    # the restatement forms coordinates / weights in float32 as OpenCV does, torch in float64: 1e-6-level differences
    assert got.shape == tuple(dst)
    assert np.abs(got - ref).max() < 5e-6


def test_identity_and_known_values():
    img = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.array_equal(ao.cv_resize(img, (3, 4), "linear"), img)
    assert np.abs(ao.cv_resize(img, (3, 4), "cubic") - img).max() < 1e-6
    up = ao.cv_resize(np.array([[0.0, 1.0]]), (1, 4), "linear")               # centres at -0.25, 0.25, 0.75, 1.25
    np.testing.assert_allclose(up[0], [0.0, 0.25, 0.75, 1.0], atol=1e-7)
    # cubic weights at t = 0.5 with A = -0.75: (-0.09375, 0.59375, 0.59375, -0.09375)
    idx, w = ao._taps(2, 1, "cubic")
    assert np.all(idx == 0)
    _, w = ao._taps(8, 4, "cubic")                                            # f = d / 2 - 0.25 -> t in {0.75, 0.25}
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-7)
    _, w = ao._taps(4, 8, "cubic")                                            # f = 2 d + 0.5 -> t = 0.5
    np.testing.assert_allclose(w[0], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)


def test_candidate_sizes_follow_the_reference_formulas():
    assert list(ao.zoom_values(64, 64, 2)) == [32, 40, 48, 56, 64]
    assert list(ao.zoom_values(48, 80, 2)) == [24, 32, 40, 48]
    rh, rw = ao.resize_sizes(64, 64, [2, 1.5])
    assert rh[0] == 32 and rh[-1] < 96 and np.array_equal(rh, rw) and len(rh) == 9       # s = 0.125: 8 / 64


def test_inter_area_enlarging_form_known_values():
    """cv2.INTER_AREA with an ENLARGED axis (what utils/img.py:cv_resize picks for small frames, predictor.py:203-204):
    OpenCV's linear kernel with area-mode coefficients = the overlap of an output pixel's footprint with the source
    pixels.  Integer factors replicate pixels (OpenCV's documented "similar to INTER_NEAREST" when zooming); 2 -> 3
    pixels gives (a, (a + b) / 2, b)."""
    row = np.array([[1.0, 5.0, 2.0]])
    up2 = ao.cv_resize(row, (1, 6), "area_up")
    np.testing.assert_allclose(up2[0], [1, 1, 5, 5, 2, 2], atol=0)
    up = ao.cv_resize(np.array([[2.0, 4.0]]), (1, 3), "area_up")
    np.testing.assert_allclose(up[0], [2.0, 3.0, 4.0], atol=1e-7)
    # footprint overlap in general: dst pixel d covers [d, d + 1) * src / dst of the source axis (<= 2 source pixels)
    rs = np.random.RandomState(0)
    src = rs.rand(1, 7)
    dst_n = 10
    got = ao.cv_resize(src, (1, dst_n), "area_up")[0]
    ref = np.zeros(dst_n)
    for d in range(dst_n):
        lo, hi = d * 7 / dst_n, (d + 1) * 7 / dst_n
        for s_ in range(7):
            ref[d] += max(0.0, min(hi, s_ + 1) - max(lo, s_)) / (hi - lo) * src[0, s_]
    np.testing.assert_allclose(got, ref, atol=2e-6)


def test_img_resize_quirks_of_the_reference():
    """utils/img.py:20-68: unequal target entries are swapped; same shape -> copy; INTER_AREA below the target size,
    INTER_CUBIC above it; float64 stack out."""
    rs = np.random.RandomState(1)
    st = rs.rand(3, 16, 16)
    same = ao.img_resize(st, (16, 16))
    assert np.array_equal(same, st) and same is not st
    up = ao.img_resize(st, (32, 32))
    assert up.shape == (3, 32, 32) and up.dtype == np.float64
    np.testing.assert_allclose(up[:, ::2, ::2], st, atol=0)              # x2 under INTER_AREA replicates pixels
    down = ao.img_resize(st, (8, 8))
    np.testing.assert_allclose(down[1], ao.cv_resize(st[1], (8, 8), "cubic"), atol=0)
    assert ao.img_resize(rs.rand(2, 8, 8), (16, 24)).shape == (2, 24, 16)  # (16, 24) is swapped to (24, 16)
    lab = ao.img_resize((st > 0.5).astype(float), (24, 24), round_=True)
    assert set(np.unique(lab)) <= {0.0, 1.0}
