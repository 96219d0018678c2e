"""Shared bodies of the checkpoint-interchange tests (reference tests:
test/models/test_loaders.py).  ``ref_*_ckpt.tar`` were written by the reference's own save_model and
``ckpt.npz`` holds what the reference predicts after its own load_model (oracle/make_golden.py ckpt)."""
import os
import warnings

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_load_reference_seg():
    import atomai_amd as aoi
    g = np.load(os.path.join(GOLD, "ckpt.npz"))
    m = aoi.models.load_model(os.path.join(GOLD, "ref_seg_unet_ckpt.tar"))
    assert isinstance(m, aoi.models.Segmentor) and not m.net.training
    assert isinstance(m.optimizer, torch.optim.Adam)
    pred = m.predict(g["seg|x"], compute_coords=False)
    assert pred.shape == g["seg|pred"].shape
    np.testing.assert_allclose(pred, g["seg|pred"], rtol=1e-4, atol=1e-6)


def check_load_reference_rvae():
    import atomai_amd as aoi
    g = np.load(os.path.join(GOLD, "ckpt.npz"))
    m = aoi.models.load_model(os.path.join(GOLD, "ref_rvae_ckpt.tar"))
    assert isinstance(m, aoi.models.rVAE) and m.translation
    zm, zs = m.encode(g["vae|x"])
    np.testing.assert_allclose(zm, g["vae|zmean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(zs, g["vae|zsd"], rtol=1e-4, atol=1e-6)
    dec = m.decode(np.array([[0.3, -0.2], [1.0, 0.5]], dtype=np.float32))
    np.testing.assert_allclose(dec, g["vae|dec"], rtol=1e-4, atol=1e-5)


def _same_optimizer(o1, o2):
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert s1["param_groups"][0]["lr"] == s2["param_groups"][0]["lr"]
    assert set(s1["state"]) == set(s2["state"]) and len(s1["state"]) > 0
    for k in s1["state"]:
        for name in ("step", "exp_avg", "exp_avg_sq"):
            a, b = s1["state"][k][name], s2["state"][k][name]
            assert np.array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy()), (k, name)


def check_roundtrip_seg(tmp_path, model):
    """test_io_segmentor + test_saved_optimizer_segmentor (test/models/test_loaders.py:62-87), plus: the file
    written here has the reference's key set and holds torch-only types."""
    import atomai_amd as aoi
    g = np.load(os.path.join(GOLD, "ckpt.npz"))
    rs = np.random.RandomState(3)
    X, Xt = rs.rand(5, 1, 8, 8), rs.rand(5, 1, 8, 8)
    y, yt = rs.randint(0, 3, (5, 8, 8)), rs.randint(0, 3, (5, 8, 8))
    seg = aoi.models.Segmentor(model, nb_classes=3, nb_filters=4)
    fname = str(tmp_path / model)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        seg.fit(X, y, Xt, yt, training_cycles=4, batch_size=2, filename=fname, plot_training_history=False)
    raw = torch.load(fname + "_metadict_final.tar", weights_only=False, map_location="cpu")
    expect = [k for k in g["seg|meta_keys"] if model == "Unet" or k != "with_dilation"]   # fcnn.py:419-442
    assert sorted(raw.keys()) == expect
    assert type(raw["optimizer"]) is torch.optim.Adam              # no atomai_amd class inside the pickle
    ref_raw = torch.load(os.path.join(GOLD, "ref_seg_unet_ckpt.tar"), weights_only=False, map_location="cpu")
    if model == "Unet":
        assert list(raw["weights"].keys()) == list(ref_raw["weights"].keys())
        for k in raw["weights"]:
            assert raw["weights"][k].shape == ref_raw["weights"][k].shape, k
            assert raw["weights"][k].dtype == ref_raw["weights"][k].dtype, k
    loaded = aoi.models.load_model(fname + "_metadict_final.tar")
    for p1, p2 in zip(loaded.net.parameters(), seg.net.parameters()):
        assert np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy())
    for (k1, b1), (k2, b2) in zip(loaded.net.named_buffers(), seg.net.named_buffers()):
        assert k1 == k2 and np.array_equal(b1.cpu().numpy(), b2.cpu().numpy()), k1
    _same_optimizer(seg.optimizer, loaded.optimizer)


def check_roundtrip_rvae(tmp_path):
    """test_io_VAE (test/models/test_loaders.py): weights and optimizer survive save -> load_model."""
    import atomai_amd as aoi
    g = np.load(os.path.join(GOLD, "ckpt.npz"))
    rs = np.random.RandomState(4)
    X = rs.rand(8, 16, 16).astype(np.float32)
    v = aoi.models.rVAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=16, numhidden_decoder=16)
    fname = str(tmp_path / "rv")
    v.fit(X, training_cycles=2, batch_size=4, filename=fname)
    raw = torch.load(fname + ".tar", weights_only=False, map_location="cpu")
    assert sorted(raw.keys()) == list(g["vae|meta_keys"])
    assert type(raw["optimizer"]) is torch.optim.Adam
    loaded = aoi.models.load_model(fname + ".tar")
    for n1, n2 in ((loaded.encoder_net, v.encoder_net), (loaded.decoder_net, v.decoder_net)):
        for p1, p2 in zip(n1.parameters(), n2.parameters()):
            assert np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy())
    _same_optimizer(v.optim, loaded.optim)
    z1, _ = v.encode(X)
    z2, _ = loaded.encode(X)
    assert np.array_equal(z1, z2)


def check_misc_loaders(tmp_path):
    import pytest
    import atomai_amd as aoi
    ref_raw = torch.load(os.path.join(GOLD, "ref_seg_unet_ckpt.tar"), weights_only=False, map_location="cpu")
    # file without 'model_type' -> state dict + warning (loaders.py:58-63)
    p = str(tmp_path / "w.tar")
    torch.save({"weights": ref_raw["weights"]}, p)
    with pytest.warns(UserWarning):
        w = aoi.models.load_model(p)
    assert list(w.keys()) == list(ref_raw["weights"].keys())
    # unknown type -> ValueError (loaders.py:55-57)
    torch.save({"model_type": "nope"}, p)
    with pytest.raises(ValueError):
        aoi.models.load_model(p)
    # ensemble: averaged weights in a single net + all members (loaders.py:236-271)
    ens = {i: {k: (v + i if v.is_floating_point() else v.clone()) for k, v in ref_raw["weights"].items()}
           for i in range(3)}
    meta = {k: v for k, v in ref_raw.items() if k not in ("weights", "optimizer")}
    meta["weights"] = ens
    torch.save(meta, p)
    net, members = aoi.models.load_ensemble(p)
    assert len(members) == 3
    sd = net.state_dict()
    for k, v in ref_raw["weights"].items():
        if k.split("_")[-1] in ("mean", "var", "tracked"):
            assert np.array_equal(sd[k].cpu().numpy(), v.numpy()), k         # BN statistics: member 0's
        else:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy() + 1.0, rtol=1e-6, atol=1e-6)
