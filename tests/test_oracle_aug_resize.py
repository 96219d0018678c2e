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
