"""Shared bodies of the Locator parity tests (emulator tier on CPU, gpu tier on the MI355X).
Integer / index work: the bar is bit-exact against the oracle and the reference golden."""
import os

import numpy as np

from oracle import locator_oracle as lo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["c3_48x64", "c1_40x40", "c2_33x47_t07", "c3_64x64_dense"]


def _same(got, want):
    assert sorted(got) == sorted(want)
    for i in want:
        assert got[i].shape == want[i].shape and got[i].dtype == np.float64, (i, got[i].shape, want[i].shape)
        assert np.array_equal(got[i], want[i]), (i, np.abs(got[i] - want[i]).max())


def check_golden(name, device):
    from atomai_amd.predictors import Locator
    g = np.load(os.path.join(GOLD, "locator.npz"))
    x = g[f"{name}|x"]
    thr, de = g[f"{name}|cfg"]
    want = {i: g[f"{name}|coords|{i}"] for i in range(len(x))}
    _same(Locator(float(thr), int(de), device=device).run(x), want)
    _same(lo.locate(x, float(thr), int(de)), want)                       # the oracle itself stays pinned
    if name == "c3_48x64":                                               # channel_first + forced 1-frame chunks
        xcf = np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))
        _same(Locator(float(thr), int(de), dim_order="channel_first", device=device, chunk_bytes=1).run(xcf), want)


def check_shapes(device, big=False):
    """Irregular components: percolation-like random masks (long snakes, holes, U-turns that need many
    union-find hooks), all-foreground, all-background, single pixels, sizes that straddle the strip /
    workgroup boundaries."""
    from atomai_amd.predictors import Locator
    rs = np.random.RandomState(17)
    shapes = [(2, 37, 53, 2, 0.45), (1, 64, 96, 3, 0.62), (3, 17, 9, 1, 0.5), (1, 130, 257, 2, 0.41)]
    if big:
        shapes += [(4, 512, 512, 3, 0.4), (2, 1024, 1024, 1, 0.41)]
    for (B, H, W, C, p) in shapes:
        x = (rs.rand(B, H, W, C) > p).astype(np.float32) * 0.9
        for de in (0, 4):
            _same(Locator(0.5, de, device=device).run(x), lo.locate(x, 0.5, de))
    ones = np.ones((1, 24, 40, 2), dtype=np.float32)
    _same(Locator(0.5, 2, device=device).run(ones), lo.locate(ones, 0.5, 2))
    zeros = np.zeros((2, 24, 40, 2), dtype=np.float32)
    got = Locator(0.5, 2, device=device).run(zeros)
    assert all(v.shape == (0, 3) for v in got.values()) and len(got) == 2
    # tile borders of the device labelling (32 x 64-pixel tiles, 8-pixel strips): everything foreground over several
    # tiles (every border pair present -> the de-duplication rule at the four-tile corners), a checkerboard (the
    # maximum number of components: every foreground pixel is a root), sides that are exact multiples of the tile,
    # a comb and a serpentine whose teeth / turns cross the tile borders (components made of many tile-local parts)
    ones_mt = np.ones((1, 70, 140, 2), dtype=np.float32)
    _same(Locator(0.5, 3, device=device).run(ones_mt), lo.locate(ones_mt, 0.5, 3))
    yy, xx = np.mgrid[0:40, 0:72]
    cb = np.zeros((1, 40, 72, 2), dtype=np.float32)
    cb[0, :, :, 0] = (yy + xx) % 2
    _same(Locator(0.5, 0, device=device).run(cb), lo.locate(cb, 0.5, 0))
    many = (rs.rand(2, 100, 200, 3) > 0.43).astype(np.float32)     # more tiles than persistent workgroups (emulator)
    _same(Locator(0.5, 2, device=device).run(many), lo.locate(many, 0.5, 2))
    onech = (rs.rand(3, 72, 136, 1) > 0.45).astype(np.float32)      # 1-channel maps, W % 8 == 0: the vector-load path
    _same(Locator(0.5, 2, device=device).run(onech), lo.locate(onech, 0.5, 2))
    exact = (rs.rand(2, 64, 128, 2) > 0.42).astype(np.float32)
    _same(Locator(0.5, 1, device=device).run(exact), lo.locate(exact, 0.5, 1))
    comb = np.zeros((1, 100, 200, 3), dtype=np.float32)
    comb[0, 97, 1:199, 0] = 1
    comb[0, 3:98, 1:199:2, 0] = 1                      # teeth two pixels apart, joined only by the bottom row
    comb[0, 2:99, 5, 1] = 1
    for k, r in enumerate(range(2, 99, 2)):            # serpentine: rows joined alternately at the right / left end
        comb[0, r, 5:195, 1] = 1
        comb[0, r + 1, 194 if k % 2 == 0 else 5, 1] = 1
    comb[0, 2:99, 5, 1] = 0
    comb[0, 2:99:2, 5, 1] = 1
    _same(Locator(0.5, 0, device=device).run(comb), lo.locate(comb, 0.5, 0))
    # spiral: one component whose first pixel is far from most of its mass
    sp = np.zeros((1, 41, 41, 2), dtype=np.float32)
    r0, r1, c0, c1 = 2, 38, 2, 38
    while r1 - r0 > 3:
        sp[0, r0, c0:c1 + 1, 0] = 1; sp[0, r0:r1 + 1, c1, 0] = 1; sp[0, r1, c0 + 2:c1 + 1, 0] = 1
        sp[0, r0 + 2:r1 + 1, c0 + 2, 0] = 1; sp[0, r0 + 2, c0 + 2:c1 - 1, 0] = 1
        r0 += 4; c0 += 4; r1 -= 4; c1 -= 4
    _same(Locator(0.5, 0, device=device).run(sp), lo.locate(sp, 0.5, 0))
    # threshold is strict (x > t) and NaN is background
    edge = np.full((1, 8, 8, 2), 0.5, dtype=np.float32)
    edge[0, 3, 3, 0] = np.nan
    edge[0, 5, 5, 0] = np.nextafter(np.float32(0.5), np.float32(1))
    got = Locator(0.5, 0, device=device).run(edge)[0]
    assert np.array_equal(got, np.array([[5.0, 5.0, 0.0]]))


def check_segmentor_predict(device):
    """Segmentor.predict(compute_coords=True): (decoded, coordinates) as the reference returns them
    (models/segmentor.py:151-200, predictors/predictor.py:262-298); coordinates must equal the oracle applied to
    the decoded maps."""
    import warnings
    import atomai_amd as aoi
    rs = np.random.RandomState(2)
    m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4)
    x = rs.rand(3, 32, 32).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        decoded, coords = m.predict(x, thresh=0.34, num_batches=2)
        decoded2 = m.predict(x, compute_coords=False)
    assert decoded.shape == (3, 32, 32, 3) and np.array_equal(decoded, decoded2)
    _same(coords, lo.locate(decoded, 0.34, 5))
    assert sum(len(v) for v in coords.values()) > 0
