"""Shared bodies of the augmentation tests (emulator tier / gpu tier)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_oracle_vs_reference_golden():
    """oracle/aug_oracle.py == the real reference's datatransform.run for the numpy / scipy steps."""
    from oracle import aug_oracle as ao
    from atomai_amd.transforms import datatransform
    g = np.load(os.path.join(GOLD, "augment.npz"))
    X = g["bb|X"]
    dt = datatransform(1, int(g["bb|seed"]), blur=[1, 50], background=True)
    P, extra = dt.draw(*X.shape)                              # the reference's scalar draws, reproduced from the seed
    ref = ao.run(X, P, blur_sigma=extra["blur_sigma"])
    np.testing.assert_allclose(ref[:, None], g["bb|out"], rtol=0, atol=1e-12)
    X1 = g["po|X"]
    dt = datatransform(1, int(g["po|seed"]), poisson_noise=[30, 40])
    P, extra = dt.draw(*X1.shape)
    xn = ao.normalize(X1)
    vals = (50 / extra["poisson_l"]) ** np.ceil(np.log2([len(np.unique(xn[0]))]))
    ref = ao.run(X1, P, fields={"poisson": g["po|draws"]}, poisson_vals=vals)
    np.testing.assert_allclose(ref[:, None], g["po|out"], rtol=0, atol=1e-12)
    # jitter: the host draws ARE the reference's shifts (scipy's poisson.rvs draws from numpy's global stream)
    Xj = g["ji|X"]
    dt = datatransform(1, int(g["ji|seed"]), jitter=[20, 50], background=True)
    P, extra = dt.draw(*Xj.shape)
    assert np.array_equal(extra["jitter"], g["ji|shifts"]) and extra["jitter"].max() > 0
    ref = ao.run(Xj, P, jitter=extra["jitter"])
    np.testing.assert_allclose(ref[:, None], g["ji|out"], rtol=0, atol=1e-12)


def check_kernels_vs_reference_golden(device):
    """The HIP path end to end (datatransform.run on the device) against the reference's output."""
    from atomai_amd.transforms import datatransform
    g = np.load(os.path.join(GOLD, "augment.npz"))
    X = torch.from_numpy(g["bb|X"]).float().to(device)
    y = torch.zeros(3, 1, 20, 28, device=device)
    out, _ = datatransform(1, int(g["bb|seed"]), blur=[1, 50], background=True).run(X, y)
    assert out.shape == g["bb|out"].shape
    assert np.abs(out.cpu().numpy() - g["bb|out"]).max() < 2e-5
    X1 = torch.from_numpy(g["po|X"]).float().to(device)
    dt = datatransform(1, int(g["po|seed"]), poisson_noise=[30, 40])
    out, _ = dt.run(X1, torch.zeros(1, 1, 24, 24, device=device),
                    fields={"poisson": torch.from_numpy(g["po|draws"])})
    assert np.abs(out.cpu().numpy() - g["po|out"]).max() < 2e-5
    Xj = torch.from_numpy(g["ji|X"]).float().to(device)
    out, _ = datatransform(1, int(g["ji|seed"]), jitter=[20, 50], background=True).run(Xj, y)
    assert np.abs(out.cpu().numpy() - g["ji|out"]).max() < 2e-5


def check_kernels_vs_oracle_all_steps(device, N=5, H=24, W=24):
    """Every step (also the skimage / cv2-owned ones, restated in the oracle) with injected noise fields."""
    from oracle import aug_oracle as ao
    from atomai_amd.transforms import datatransform
    rs = np.random.RandomState(0)
    X = rs.rand(N, H, W)
    lab = rs.randint(0, 3, (N, H, W))
    fields = {"gauss": rs.randn(N, H, W), "sp_flip": rs.rand(N, H, W), "sp_salt": rs.rand(N, H, W)}
    dt = datatransform(3, 5, rotation=True, gauss_noise=True, jitter=[10, 60], salt_and_pepper=[20, 50], contrast=True,
                       background=True)
    out, tl = dt.run(torch.from_numpy(X).float().to(device), torch.from_numpy(lab).to(device),
                     fields={k: torch.from_numpy(v) for k, v in fields.items()})
    P = dt.params
    jit = dt.extra["jitter"]
    assert jit.shape == (N, H) and jit.max() > 0
    assert set(P[:, 0]) <= {-1, 0, 1, 2} and (P[:, 1] > 0).any() and (P[:, 4] > 0).all()
    # rotated noise fields: the kernel indexes its fields by OUTPUT pixel, as the reference applies noise after rotating
    ref = ao.run(X, P, fields=fields, jitter=jit)             # (labels are not jittered: imaug.py:135)
    assert out.shape == (N, 1, H, W)
    assert np.abs(out[:, 0].cpu().numpy() - ref).max() < 3e-5
    for i in range(N):
        assert np.array_equal(tl[i].cpu().numpy(), ao.flip(lab[i], int(P[i, 0])))


def check_generator_statistics(device, N=4, H=64, W=64):
    """The in-kernel Philox sampler: gaussian mean / variance, poisson mean / variance (small and large rates), salt &
    pepper fractions; determinism for equal seeds, different fields for different seeds."""
    from atomai_amd import _lib as L
    x = torch.full((N, H, W), 0.5, device=device)
    P = np.zeros((N, 12), np.float32)
    P[:, 0] = 4

    def point(P_, seed):
        y = torch.empty_like(x)
        Pd = torch.from_numpy(P_).to(device)
        L.call("amx_aug_point", L.ptr(x), L.ptr(y), L.ptr(Pd), None, None, None, None, None, None, N, H, W, seed,
               L.stream_ptr(x))
        return y.cpu().numpy()
    Pg = P.copy(); Pg[:, 1] = 0.05
    a, b, c = point(Pg, 1), point(Pg, 1), point(Pg, 2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    z = (a - 0.5) / 0.05
    assert abs(z.mean()) < 0.03 and abs(z.var() - 1) < 0.05
    assert abs(np.corrcoef(z[0].ravel(), z[1].ravel())[0, 1]) < 0.05          # images get independent streams
    for scale in (6.0, 400.0):                                                # lambda = 3 and 200
        Pp = P.copy(); Pp[:, 2] = scale
        k = point(Pp, 3) * scale
        lam = 0.5 * scale
        assert np.allclose(k, np.round(k), atol=1e-3)
        assert abs(k.mean() - lam) < 0.04 * lam and abs(k.var() - lam) < 0.08 * lam
    Ps = P.copy(); Ps[:, 3] = 0.2
    s = point(Ps, 4)
    salt, pepper = (s == 1).mean(), (s == 0).mean()
    assert abs(salt - 0.1) < 0.01 and abs(pepper - 0.1) < 0.01


def check_class_drop_and_trainer_hook(device):
    """Pairs in which a class is absent are dropped (squeeze_channels); seg_augmentor plugs into the trainer."""
    import atomai_amd as aoi
    from atomai_amd.transforms import datatransform, seg_augmentor
    rs = np.random.RandomState(1)
    X = torch.from_numpy(rs.rand(4, 16, 16).astype(np.float32)).to(device)
    lab = rs.randint(0, 3, (4, 16, 16))
    lab[2][lab[2] == 1] = 0                                   # image 2 has no class 1
    out, tl = datatransform(3, 0, rotation=True).run(X, torch.from_numpy(lab).to(device))
    assert out.shape == (3, 1, 16, 16) and tl.shape == (3, 16, 16) and tl.dtype == torch.int64
    assert float(out.min()) == 0.0 and float(out.max()) == 1.0
    assert seg_augmentor(3) is None
    check_custom_transform(device)
    Xn = rs.rand(8, 16, 16).astype(np.float32)
    yn = rs.randint(0, 3, (8, 16, 16))
    m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4, seed=1)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:               # fit() writes <filename>_metadict_final.tar
        m.fit(Xn, yn, Xn[:4], yn[:4], training_cycles=3, batch_size=4, rotation=True, gauss_noise=[20, 40], contrast=True,
              background=True, plot_training_history=False, filename=os.path.join(tmp, "model"))
    assert len(m.loss_acc["train_loss"]) == 3 and all(np.isfinite(m.loss_acc["train_loss"]))


def check_custom_transform(device):
    """``custom_transform`` (reference imaug.py:79, 323-324): the user's host callable sees the normalised images
    (N, H, W) float64 and the one-hot masks (N, H, W, C) float64 and runs FIRST; the device steps follow without a second
    normalisation.  Checked through equivalences that need no cv2 / skimage: the identity callable changes nothing, a
    flip inside the callable equals flipping the inputs (min / max normalisation commutes with it), a callable may drop
    images, and what it receives is what the reference would pass."""
    from atomai_amd.transforms import datatransform, seg_augmentor
    rs = np.random.RandomState(3)
    X = torch.from_numpy((rs.rand(5, 16, 24) * 3 + 1).astype(np.float32)).to(device)
    lab = torch.from_numpy(rs.randint(0, 3, (5, 16, 24))).to(device)
    kw = dict(rotation=True, gauss_noise=[20, 40], contrast=True, background=True)
    seen = {}

    def ident(a, b):
        seen["a"], seen["b"] = a.copy(), b.copy()
        return a, b
    base = datatransform(3, 7, **kw).run(X, lab)
    got = datatransform(3, 7, custom_transform=ident, **kw).run(X, lab)
    assert torch.equal(base[0], got[0]) and torch.equal(base[1], got[1])
    xn = X.cpu().numpy()
    assert seen["a"].dtype == np.float64 and seen["a"].shape == (5, 16, 24) and seen["b"].shape == (5, 16, 24, 3)
    np.testing.assert_allclose(seen["a"], (xn - xn.min()) / np.ptp(xn), rtol=0, atol=2e-7)
    assert np.array_equal(seen["b"], np.eye(3)[lab.cpu().numpy()])
    flip = lambda a, b: (a[:, :, ::-1], b[:, :, ::-1])                    # noqa: E731
    got = datatransform(3, 7, custom_transform=flip, **kw).run(X, lab)
    ref = datatransform(3, 7, **kw).run(X.flip(2).contiguous(), lab.flip(2).contiguous())
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    got = datatransform(3, 7, custom_transform=lambda a, b: (a[1:] * 0.5, b[1:]), **kw).run(X, lab)
    assert got[0].shape[0] <= 4 and got[0].shape[1:] == (1, 16, 24) and float(got[0].max()) == 1.0
    # binary masks (one class): (N, 1, H, W) float in, the callable sees (N, H, W, 1)
    Xb = X[:, :16, :16].contiguous()
    mb = (torch.from_numpy(rs.rand(5, 1, 16, 16)) > 0.5).float().to(device)
    got = datatransform(1, 2, custom_transform=lambda a, b: (a, 1.0 - b), rotation=True).run(Xb, mb)
    ref = datatransform(1, 2, rotation=True).run(Xb, 1.0 - mb)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and got[1].shape == (5, 1, 16, 16)
    # masks that are not exactly one-hot: the reference never raises — it carries them through its
    # geometric steps and squeeze_channels decides per frame.  Here they are rounded and squeezed on arrival:
    soft = lambda a, b: (a, 0.9 * b + 0.04)                               # noqa: E731  (rounds back to the one-hot masks)
    got = datatransform(3, 7, custom_transform=soft, **kw).run(X, lab)
    assert torch.equal(base[0], got[0]) and torch.equal(base[1], got[1])

    def overlap_first(a, b):                                  # frame 0: a patch where every channel is set (label 3 > 2)
        b = b.copy()
        b[0, :4, :4, :] = 1.0
        return a, b
    got = datatransform(3, 7, custom_transform=overlap_first, rotation=True).run(X, lab)
    assert got[0].shape[0] <= 4                                # that frame is dropped, the batch goes on
    try:
        datatransform(3, 0, custom_transform=lambda a, b: (a, b * 0.0 + 1.0), rotation=True).run(X, lab)
        raise AssertionError("a batch whose every frame is unusable must say so")
    except RuntimeError as e:
        assert "custom_transform left no usable frame" in str(e)
    aug = seg_augmentor(3, custom_transform=ident, rotation=True)
    assert aug is not None


def check_img_resize(device):
    """utils.img_resize / SegPredictor(resize=...) (reference utils/img.py:20-68, predictors/predictor.py:203-204) on
    amx_aug_resample against the numpy restatement oracle/aug_oracle.py:img_resize (both UNPINNED against cv2 itself):
    enlarging (INTER_AREA, mode 2), shrinking (INTER_CUBIC), the swapped non-square target, rounding, same-shape copy."""
    from oracle import aug_oracle as ao
    import atomai_amd as aoi
    from atomai_amd.utils import img_resize
    rs = np.random.RandomState(5)
    st = rs.rand(3, 20, 20).astype(np.float32)
    for target in ((32, 32), (40, 40), (27, 27), (12, 12), (16, 24)):
        got = img_resize(st, target)
        ref = ao.img_resize(st.astype(np.float64), target)
        assert got.shape == ref.shape and got.dtype == np.float64, target
        assert np.abs(got - ref).max() < 5e-6, (target, np.abs(got - ref).max())
    assert np.array_equal(img_resize(st, (40, 40))[:, ::2, ::2], st.astype(np.float64))   # x2 INTER_AREA replicates
    same = img_resize(st, (20, 20))
    assert np.array_equal(same, st) and same is not st
    lab = img_resize((st > 0.5).astype(np.float32), (28, 28), round_=True)
    assert np.array_equal(lab, ao.img_resize((st > 0.5).astype(np.float64), (28, 28), round_=True))
    # the predictor resizes before padding / normalising (predictor.py:203-206): outputs come back at the new size
    torch.manual_seed(0)
    net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=4)
    p = aoi.predictors.SegPredictor(net, resize=(32, 32), use_gpu=(device != "cpu"), verbose=False)
    out = p.predict(st, compute_coords=False)
    assert out.shape == (3, 32, 32, 3)
    p0 = aoi.predictors.SegPredictor(net, use_gpu=(device != "cpu"), verbose=False)
    ref = p0.predict(img_resize(st, (32, 32)).astype(np.float32), compute_coords=False)
    assert np.abs(out - ref).max() < 1e-6


def check_augment_geometry_golden(device):
    """rotation -> zoom -> resize against tests/golden/augment_geom.npz: the reference's own seg_augmentor / datatransform
    code run over the documented cv2 semantics (oracle/make_golden.py augment_geom).  Same seed -> the reference's zoom
    windows / output size / flips; class maps must agree EXACTLY (incl. which pairs are dropped), images to fp32 accuracy."""
    import ast
    import os
    from atomai_amd.transforms import seg_augmentor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_geom.npz"))
    names = sorted({k.split("|")[0] for k in g.files})
    assert len(names) == 6
    for name in names:
        K, seed = (int(v) for v in g[f"{name}|cfg"])
        kw = ast.literal_eval(str(g[f"{name}|kw"]))
        aug = seg_augmentor(K, **kw)
        x = torch.from_numpy(g[f"{name}|x"]).to(device)
        lab = torch.from_numpy(g[f"{name}|lab"]).to(device)
        xo, lo = aug(x, lab, seed)
        rx, rl = g[f"{name}|x_out"], g[f"{name}|lab_out"]
        assert tuple(xo.shape) == rx.shape and tuple(lo.shape) == rl.shape, (name, xo.shape, rx.shape, lo.shape, rl.shape)
        lo_np = lo.cpu().numpy()
        if K > 1:
            assert lo.dtype == torch.int64
            mism = (lo_np != rl).mean()
            # a one-hot plane interpolates to exactly 0.5 only on measure-zero inputs; fp32 vs fp64 weights may flip a
            # handful of pixels whose interpolated mask value lies within 1e-6 of 0.5
            assert mism <= 2e-4, (name, mism)
        else:
            assert (lo_np != rl).mean() <= 2e-4, name
        assert np.abs(xo.cpu().numpy() - rx).max() < 5e-5, (name, np.abs(xo.cpu().numpy() - rx).max())
