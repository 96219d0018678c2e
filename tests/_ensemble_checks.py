"""Shared body of the EnsembleTrainer parity test (reference tests:
test/trainers/test_etrainer.py).  Golden: tests/golden/ensemble.npz (oracle/make_golden.py ensemble)."""
import os
import warnings

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


TOL = [5e-3]      # relative bound of _close (set per tier by the callers of check_ensemble)


def _close(sd, g, prefix, xt):
    """Member weights after a few fp32 Adam steps are not comparable tensor by tensor: Adam normalises the gradient,
    so every weight whose true gradient is (near) zero — conv biases in front of a BatchNorm, the 2x2-pixel
    bottleneck — moves by up to lr per step in a direction set by rounding noise.  The members are compared as
    FUNCTIONS instead: eval-mode logits of the reference's weights vs ours on the test images (same criterion as the
    3-step eval check of the net-level goldens), plus exact equality of the integer buffers."""
    import atomai_amd as aoi
    from collections import OrderedDict
    ref_sd = OrderedDict((k, torch.from_numpy(g[f"{prefix}|{k}"])) for k in sd)
    for k, v in sd.items():
        if "num_batches_tracked" in k:
            assert np.array_equal(v.cpu().numpy(), ref_sd[k].numpy()), k
    outs = []
    for weights in (ref_sd, sd):
        net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=4)
        net.load_state_dict(weights)
        net.to(next(iter(sd.values())).device).eval()
        with torch.no_grad():
            outs.append(net(xt.to(next(net.parameters()).device)).cpu().numpy())
    err = np.abs(outs[0] - outs[1]).max() / np.abs(outs[0]).max()
    assert err < TOL[0], (prefix, err, TOL[0])


def check_ensemble(tmp_path, tol=5e-3):
    import atomai_amd as aoi
    TOL[0] = tol
    g = np.load(os.path.join(GOLD, "ensemble.npz"))
    X, y, Xt, yt = g["X"], g["y"], g["Xt"], g["yt"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(training_cycles=3, batch_size=2, plot_training_history=False,
                                    filename=str(tmp_path / "ens_a"))
        net, ens = et.train_ensemble_from_scratch(X, y, Xt, yt, n_models=2)
    assert sorted(ens) == [0, 1]
    xt = torch.from_numpy(Xt[:, None])
    for i in ens:
        _close(ens[i], g, f"scratch|{i}", xt)
    np.testing.assert_allclose(et.loss_acc["train_loss"], g["scratch|last_train_loss"], rtol=1e-4)
    # members differ from one another (test_etrainer.py: not any(m_not_eq))
    assert any(not torch.equal(ens[0][k], ens[1][k]) for k in ens[0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(batch_size=2, plot_training_history=False, filename=str(tmp_path / "ens_b"))
        net, ens = et.train_ensemble_from_baseline(X, y, Xt, yt, n_models=2, training_cycles_base=3,
                                                   training_cycles_ensemble=2)
    for i in ens:
        _close(ens[i], g, f"baseline|{i}", xt)
    _close(net.state_dict(), g, "baseline|avg", xt)
    raw = torch.load(str(tmp_path / "ens_b") + "_ensemble_metadict.tar", weights_only=False, map_location="cpu")
    assert sorted(raw.keys()) == list(g["ens_meta_keys"])
    smodel, members = aoi.models.load_ensemble(str(tmp_path / "ens_b") + "_ensemble_metadict.tar")
    assert len(members) == 2
    for k, v in smodel.state_dict().items():
        if k.split("_")[-1] not in ("mean", "var", "tracked"):
            np.testing.assert_allclose(v.cpu().numpy(), net.state_dict()[k].cpu().numpy(), rtol=1e-6, atol=1e-7)


def check_ensemble_predictor():
    """EnsemblePredictor.predict on the REFERENCE-trained members (golden weights): mean and variance of the class
    probabilities over the members (predictors/epredictor.py:121-295)."""
    import atomai_amd as aoi
    from collections import OrderedDict
    g = np.load(os.path.join(GOLD, "ensemble.npz"))
    net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=4)
    ens = {i: OrderedDict((k, torch.from_numpy(g[f"scratch|{i}|{k}"])) for k in net.state_dict()) for i in (0, 1)}
    ep = aoi.predictors.EnsemblePredictor(net, ens, nb_classes=3, verbose=0)
    mean, var = ep.predict(g["Xt"], num_batches=2)
    assert mean.shape == g["epred|mean"].shape == (4, 16, 16, 3) and mean.dtype == np.float64
    np.testing.assert_allclose(mean, g["epred|mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(var, g["epred|var"], rtol=2e-3, atol=1e-9)
    mean_cf, _ = ep.predict(g["Xt"], num_batches=1, format_out="channel_first")
    np.testing.assert_allclose(mean_cf.transpose(0, 2, 3, 1), mean, rtol=1e-6, atol=1e-9)
    # ensemble_locate: members x frames x H x W x C -> per-frame cluster means (DBSCAN over the members' centres)
    rs = np.random.RandomState(3)
    from oracle import locator_oracle as lo
    frame = lo.synthetic_maps(rs, 1, 48, 48, 2, n_blobs=8)[0]
    stack = np.stack([np.roll(frame, s, axis=1) for s in (0, 0, 0, 0)] * 3)[:, None]      # 12 identical members
    cm, cv = aoi.predictors.ensemble_locate(stack, eps=0.5, threshold=0.5)
    ref = lo.locate(frame[None], 0.5, 5)[0]
    rows = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]                                      # noqa: E731
    # reference quirk kept on purpose: cluster_coord skips np.unique(labels)[0], assuming it is DBSCAN's noise
    # label -1; with no noise points it is cluster 0 (= the first detected centre) that is dropped
    assert cm[0].shape == (len(ref) - 1, 2), (cm[0].shape, ref.shape)
    assert np.allclose(rows(cm[0]), rows(ref[1:, :2]))
    assert np.allclose(cv[0], 0)
