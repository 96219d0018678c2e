"""The reference's OWN test suite for the hot path, restated against atomai_amd (what a user of the reference would run
after switching the import).  Scenarios and expected values follow /root/reference/test — trainers/test_trainer.py
(Seg* tests), trainers/test_vitrainer.py, models/test_vae.py (VAE / rVAE tests), predictors/test_predictor.py
(Base / SegPredictor tests), transforms/test_imaug.py — with the same dummy-data shapes; the bodies are written for this
repo's test tiers (``dev`` = "cpu" under the SIMT emulator, "cuda" on the MI355X).  Families outside SURVEY §8 (ImSpec,
Reg / cls trainers, joint VAEs, custom torchvision backbones) are not part of it."""
import numpy as np
import torch


def _images(n=5):
    rs = np.random.RandomState(0)
    return rs.random_sample((n, 1, 8, 8)), rs.random_sample((n, 1, 8, 8))


def _labels(binary, n=5):
    rs = np.random.RandomState(1)
    if binary:
        return rs.randint(0, 2, (n, 1, 8, 8)), rs.randint(0, 2, (n, 1, 8, 8))
    return rs.randint(0, 3, (n, 8, 8)), rs.randint(0, 3, (n, 8, 8))


def _trainer(model="Unet", binary=False, cycles=1, **kw):
    from atomai_amd.trainers import SegTrainer
    X, Xt = _images()
    y, yt = _labels(binary)
    t = SegTrainer(model, nb_classes=1 if binary else 3, **kw)
    t.compile_trainer((X, y, Xt, yt), training_cycles=cycles, batch_size=4, plot_training_history=False)
    return t


# ------------------------------------------------------------------ trainers/test_trainer.py
def loss_selection():
    assert str(_trainer(binary=True).criterion) == "BCEWithLogitsLoss()"
    assert str(_trainer(binary=False).criterion) == "CrossEntropyLoss()"


def segtrainer_determinism(model_type, tmp, cycles=5, **kw):
    """(``kw``: the emulator tier trains narrow nets — nb_filters=4 — the gpu tier the reference's default widths)"""
    out = []
    for _ in range(2):
        from atomai_amd.trainers import SegTrainer
        X, Xt = _images()
        y, yt = _labels(True)
        t = SegTrainer(model_type, upsampling="nearest", seed=1, **kw)
        t.compile_trainer((X, y, Xt, yt), training_cycles=cycles, batch_size=4, plot_training_history=False,
                          filename=str(tmp / "m"))
        t.run()
        out.append((t.loss_acc["train_loss"][-1], [p.detach().cpu().numpy().copy() for p in t.net.parameters()]))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)                      # the reference asserts allclose; this path is bit-deterministic


def segtrainer_dataloader(binary, dev):
    t = _trainer(binary=binary)
    X_, y_ = t.dataloader(0)
    assert X_.dtype == torch.float32
    assert y_.dtype == (torch.float32 if binary else torch.int64)
    assert X_.is_cuda == (dev == "cuda")


def init_unet_bn(bn, n_bn, layers):
    t = _trainer("Unet", batch_norm=bn, layers=layers)
    assert len([k for k in t.net.state_dict() if "running_mean" in k]) == n_bn


def init_dropouts(model, dropout, expected):
    t = _trainer(model, dropout=dropout)
    n = sum(1 for c in t.net.children() if "Dropout" in str([m for m in c.named_modules()]))
    assert n == expected


def init_unet_layers(layers):
    t = _trainer("Unet", layers=layers)
    keys = list(t.net.state_dict())
    n_bn = len([k for k in keys if "running_mean" in k])
    assert len([k for k in keys if "weight" in k]) - 4 - n_bn == 2 * sum(layers[:-1]) + layers[-1]


def init_dilnet_bn(bn, n_bn, layers):
    t = _trainer("dilnet", batch_norm=bn, layers=layers)
    assert len([k for k in t.net.state_dict() if "running_mean" in k]) == n_bn


def init_dilnet_layers(layers):
    t = _trainer("dilnet", layers=layers)
    keys = list(t.net.state_dict())
    n_bn = len([k for k in keys if "running_mean" in k])
    assert len([k for k in keys if "weight" in k]) - 2 - n_bn == sum(layers)


def init_segmodel_filters(model, nb_filters, expected):
    from atomai_amd.nets.blocks import UpsampleBlock
    t = _trainer(model, batch_norm=False, nb_filters=nb_filters)
    got = []
    for child in t.net.children():
        if isinstance(child, UpsampleBlock):
            continue
        got.append(np.unique([p.shape[0] for p in child.state_dict().values()])[0])
    assert list(got[:-1]) == expected


# ------------------------------------------------------------------ trainers/test_vitrainer.py
def vi_set_nets(enc_name, dec_name, separately):
    import atomai_amd.nets as nets
    from atomai_amd.trainers import viBaseTrainer
    enc, dec = getattr(nets, enc_name)((28, 28), 2), getattr(nets, dec_name)((28, 28), 2)
    v = viBaseTrainer()
    if separately:
        v.set_encoder(enc), v.set_decoder(dec)
    else:
        v.set_model(enc, dec)
    assert hasattr(v.encoder_net, "state_dict") and hasattr(v.decoder_net, "state_dict")


def vi_set_data(torch_format):
    from atomai_amd.trainers import viBaseTrainer
    X = np.random.RandomState(0).random_sample((100, 28, 28))
    v = viBaseTrainer()
    v.set_data(torch.from_numpy(X).float() if torch_format else X)
    assert isinstance(v.train_iterator, torch.utils.data.DataLoader)


def vi_reparametrize(dev):
    from atomai_amd.nets import fcEncoderNet
    from atomai_amd.trainers import viBaseTrainer
    X = torch.from_numpy(np.random.RandomState(0).random_sample((100, 28, 28))).float().to(dev)
    v = viBaseTrainer()
    v.set_encoder(fcEncoderNet((28, 28), 2))
    z_mu, z_sd = v.encoder_net(X)
    z = v.reparameterize(z_mu, z_sd)
    assert z.shape == (100, 2) and not torch.equal(z, z_mu)


def vi_custom_optimizer():
    """Two learning rates give different training curves (the reference trains 2 epochs with each)."""
    import atomai_amd as aoi
    X = np.random.RandomState(0).random_sample((32, 12, 12)).astype(np.float32)
    losses = []
    for lr in (1e-2, 1e-6):
        m = aoi.models.VAE((12, 12), latent_dim=2, seed=0, numhidden_encoder=16, numhidden_decoder=16)
        m.compile_trainer((X, None), None, optimizer=lambda p, lr=lr: torch.optim.Adam(p, lr=lr), training_cycles=2,
                          batch_size=8)
        assert isinstance(m.optim, torch.optim.Adam) and m.optim.param_groups[0]["lr"] == lr
        for e in range(2):
            m.loss_history["train_loss"].append(m.train_epoch())
        losses.append(m.loss_history["train_loss"][-1])
    assert losses[0] != losses[1]


# ------------------------------------------------------------------ models/test_vae.py
def _vae_data(n=12, hw=(8, 8)):
    return np.random.RandomState(2).random_sample((n,) + hw).astype(np.float32)


def vae_encoding(kind, latent_dim, translation=True):
    import atomai_amd as aoi
    X = _vae_data()
    if kind == "VAE":
        m = aoi.models.VAE((8, 8), latent_dim=latent_dim, numhidden_encoder=16, numhidden_decoder=16)
        want = latent_dim
    else:
        m = aoi.models.rVAE((8, 8), latent_dim=latent_dim, translation=translation, numhidden_encoder=16,
                            numhidden_decoder=16)
        want = latent_dim + (3 if translation else 1)
    z_mean, z_sd = m.encode(X)
    assert z_mean.shape == z_sd.shape == (len(X), want)


def vae_decoding(kind, conv_encoder, conv_decoder, latent_dim, translation=True, nb_classes=0):
    import atomai_amd as aoi
    kw = dict(conv_encoder=conv_encoder, numhidden_encoder=16, numhidden_decoder=16, nb_classes=nb_classes)
    if kind == "VAE":
        m = aoi.models.VAE((8, 8), latent_dim=latent_dim, conv_decoder=conv_decoder, **kw)
    else:
        m = aoi.models.rVAE((8, 8), latent_dim=latent_dim, translation=translation, **kw)
    z = np.random.RandomState(3).randn(latent_dim).astype(np.float32)
    out = m.decode(z, 1) if nb_classes else m.decode(z)
    assert out.shape == (1, 8, 8)


def vae_reconstruct(conv_encoder, conv_decoder, latent_dim):
    import atomai_amd as aoi
    m = aoi.models.VAE((8, 8), latent_dim=latent_dim, conv_encoder=conv_encoder, conv_decoder=conv_decoder,
                       numhidden_encoder=16, numhidden_decoder=16)
    out = m.reconstruct(_vae_data(1)[0], num_samples=3)
    assert out.shape == (3, 8, 8)


def vae_encode_image(kind, latent_dim):
    import atomai_amd as aoi
    cls = aoi.models.VAE if kind == "VAE" else aoi.models.rVAE
    m = cls((8, 8), latent_dim=latent_dim, numhidden_encoder=16, numhidden_decoder=16)
    img = np.random.RandomState(4).random_sample((16, 16)).astype(np.float32)
    img_, enc = m.encode_image_(img, num_batches=2)
    zdim = latent_dim if kind == "VAE" else latent_dim + 3
    assert enc.shape[:2] == img_.shape and enc.shape[-1] == zdim


# ------------------------------------------------------------------ predictors/test_predictor.py
def basepredictor(dev):
    from atomai_amd.predictors import BasePredictor
    net = torch.nn.Sequential(torch.nn.Linear(8, 4))
    p = BasePredictor(net, use_gpu=(dev == "cuda"))
    x_np = np.random.RandomState(5).randn(5, 8)
    assert isinstance(p.preprocess(x_np), torch.Tensor) and p.preprocess(x_np).dtype == torch.float32
    xt = torch.randn(5, 8)
    assert p.preprocess(xt) is xt
    assert p.forward_(xt).shape == (5, 4)
    for bs in (1, 2, 5):
        assert p.batch_predict(xt, (5, 4), bs).shape == (5, 4)
    assert p.predict(xt, (4,), num_batches=2).shape == (5, 4)
    assert next(p.model.parameters()).is_cuda == (dev == "cuda")


def segpredictor(model, shape, dev):
    import atomai_amd as aoi
    torch.manual_seed(0)
    net, _ = aoi.nets.init_fcnn_model(model, 1, **({"nb_filters": 4} if model in ("Unet", "dilnet", "SegResNet") else {"nb_filters": 4}))
    x = np.random.RandomState(6).random_sample(shape).astype(np.float32)
    p = aoi.predictors.SegPredictor(net, use_gpu=(dev == "cuda"), nb_classes=1, verbose=False)
    assert p.predict(x).shape == (2 if len(shape) == 3 else 1, 8, 8, 1)
    dec = p.run(x, compute_coords=False)
    assert dec.shape == (2 if len(shape) == 3 else 1, 8, 8, 1)
    dec2, coords = p.run(x, compute_coords=True)
    assert np.array_equal(dec, dec2) and isinstance(coords, dict) and len(coords) == dec.shape[0]
    for v in coords.values():
        assert v.ndim == 2 and v.shape[1] == 3


# ------------------------------------------------------------------ transforms/test_imaug.py
def imaug_individual(kw, dev):
    from atomai_amd.transforms import datatransform
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.random_sample((6, 32, 32)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(((np.mgrid[0:32, 0:32][0][None] // 4 + rs.randint(0, 3, (6, 1, 1))) % 3).astype(np.int64)).to(dev)
    xo, lo = datatransform(3, seed=1, **kw).run(x, lab)
    assert xo.ndim == 4 and xo.shape[1] == 1 and lo.shape[0] == xo.shape[0] and lo.shape[1:] == xo.shape[2:]
    assert abs(float(xo.min())) < 1e-6 and abs(float(xo.max()) - 1.0) < 1e-6
    if not any(k in kw for k in ("zoom", "resize")):
        assert tuple(xo.shape[2:]) == (32, 32)
    if not any(k in kw for k in ("zoom", "resize", "rotation")):
        assert torch.equal(lo, lab)                          # noise transforms leave the labels alone
    assert not torch.equal(xo[:, 0], x)


# ------------------------------------------------------------------ models/test_dklgpr.py, trainers/test_gptrainer.py
def _dkl_data(ydim=(50,), indim=32):
    rs = np.random.RandomState(8)
    return rs.randn(50, indim), rs.randn(*ydim), rs.randn(50, indim)


def dkl_fit():
    import atomai_amd as aoi
    X, y, _ = _dkl_data()
    t = aoi.models.dklGPR(32, precision="single")
    assert len(t.train_loss) == 0
    t.fit(X, y, 2)
    assert len(t.train_loss) == 2


def dkl_fit_ensemble(shared_emb):
    import atomai_amd as aoi
    X, y, _ = _dkl_data()
    t = aoi.models.dklGPR(32, precision="single", shared_embedding_space=shared_emb)
    t.fit_ensemble(X, y, 2, n_models=3)
    assert len(t.train_loss) == 2 and len(t.gp_model.models) == 3
    w = [m.feature_extractor.linear1.weight.detach().cpu() for m in t.gp_model.models]
    assert not torch.equal(w[0], w[1])                       # every member has its own initialisation


def dkl_predict():
    import atomai_amd as aoi
    X, y, Xt = _dkl_data()
    t = aoi.models.dklGPR(32, precision="single")
    t.fit(X, y)
    mean, var = t.predict(Xt)
    assert isinstance(mean, np.ndarray) and isinstance(var, np.ndarray) and mean.shape == var.shape == (50,)


def dkl_multi_model_predict():
    import atomai_amd as aoi
    X, y, Xt = _dkl_data((2, 50))
    t = aoi.models.dklGPR(32, shared_embedding_space=False, precision="single")
    t.fit(X, y)
    mean, var = t.predict(Xt)
    assert mean.shape == var.shape == (2, 50)


def dkl_ensemble_predict(shared_emb, ydim):
    import atomai_amd as aoi
    X, y, Xt = _dkl_data(ydim)
    t = aoi.models.dklGPR(32, shared_embedding_space=shared_emb, precision="single")
    t.fit_ensemble(X, y, 1, n_models=3)
    mean, var = t.predict(Xt)
    assert mean.shape == var.shape == (3, 50)


def dkl_sampling(reg_dim, shared=True):
    import atomai_amd as aoi
    X, y, Xt = _dkl_data((reg_dim, 50))
    t = aoi.models.dklGPR(32, precision="single", shared_embedding_space=shared)
    t.fit(X, y)
    s = t.sample_from_posterior(Xt, 100)
    assert isinstance(s, np.ndarray) and s.shape == (100, reg_dim, 50)
    sample, xnext = t.thompson(Xt)
    assert isinstance(sample, np.ndarray) and isinstance(xnext, np.ndarray) and sample.shape == (reg_dim, 50)


# ------------------------------------------------------------------ trainers/test_gptrainer.py (dklGPTrainer tests)
def _state_equal(a, b):
    return all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)


def dkltrainer_precision(precision, dtype):
    from atomai_amd.trainers import dklGPTrainer
    X, y, _ = _dkl_data()
    X_, y_ = dklGPTrainer(32, precision=precision).set_data(X, y)
    assert X_.dtype == dtype and y_.dtype == dtype


def dkltrainer_compile_train_run(tmp):
    import copy
    from atomai_amd.trainers import dklGPTrainer
    X, y, _ = _dkl_data()
    t = dklGPTrainer(32, precision="single")
    t.compile_trainer(X, y, 2)
    assert t.gp_model is not None and t.likelihood is not None
    w0 = copy.deepcopy(t.gp_model.feature_extractor.state_dict())
    t.train_step()
    assert not _state_equal(w0, t.gp_model.feature_extractor.state_dict())
    t = dklGPTrainer(32, precision="single")                   # frozen extractor: transfer learning
    t.compile_trainer(X, y, freeze_weights=True)
    w0 = copy.deepcopy(t.gp_model.feature_extractor.state_dict())
    t.train_step()
    assert _state_equal(w0, t.gp_model.feature_extractor.state_dict())
    t = dklGPTrainer(32, precision="single")
    t.compile_trainer(X, y, training_cycles=3)
    t.run()
    assert len(t.train_loss) == 3
    t = dklGPTrainer(32, precision="single")
    t.run(X, y, 3)
    assert len(t.train_loss) == 3
    w1 = copy.deepcopy(t.gp_model.feature_extractor.state_dict())
    t.save_weights(str(tmp / "m.pt"))
    t.gp_model.feature_extractor.load_state_dict(torch.load(str(tmp / "m.pt")))
    assert _state_equal(w1, t.gp_model.feature_extractor.state_dict())


def dkltrainer_multi_model():
    import copy
    from atomai_amd.trainers import dklGPTrainer
    X, y, _ = _dkl_data((2, 50))
    t = dklGPTrainer(32, shared_embedding_space=False, precision="single")
    t.compile_multi_model_trainer(X, y, 2)
    assert t.gp_model is not None and t.likelihood is not None
    w1 = copy.deepcopy(t.gp_model.models[0].feature_extractor.state_dict())
    w2 = copy.deepcopy(t.gp_model.models[1].feature_extractor.state_dict())
    assert _state_equal(w1, w2)                              # independent outputs start from the same initial network
    t.train_step()
    f1, f2 = t.gp_model.models[0].feature_extractor.state_dict(), t.gp_model.models[1].feature_extractor.state_dict()
    assert not _state_equal(f1, f2) and not _state_equal(w1, f1) and not _state_equal(w2, f2)
    t = dklGPTrainer(32, precision="single", shared_embedding_space=False)     # ensemble: own initialisation each
    t.ensemble = True
    t.compile_multi_model_trainer(X, np.repeat(y[:1], 3, axis=0))
    assert not _state_equal(t.gp_model.models[0].state_dict(), t.gp_model.models[2].state_dict())
    try:
        dklGPTrainer(32, precision="single").compile_multi_model_trainer(X, y)
        raise AssertionError("shared embedding space must refuse compile_multi_model_trainer")
    except NotImplementedError:
        pass


# ------------------------------------------------------------------ models/test_loaders.py
def _opt_equal(o1, o2):
    for g1, g2 in zip(o1.param_groups, o2.param_groups):
        for p1, p2 in zip(g1["params"], g2["params"]):
            if not np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy()):
                return False
    return True


def io_segmentor(model, tmp, **kw):
    import atomai_amd as aoi
    X, Xt = _images()
    y, yt = _labels(False)
    seg = aoi.models.Segmentor(model, nb_classes=3, **kw)
    seg.fit(X, y, Xt, yt, training_cycles=4, batch_size=2, filename=str(tmp / model), plot_training_history=False)
    loaded = aoi.models.load_model(str(tmp / f"{model}_metadict_final.tar"))
    for p1, p2 in zip(loaded.net.parameters(), seg.net.parameters()):
        assert np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy())
    assert _opt_equal(seg.optimizer, loaded.optimizer)


def io_vae(kind, tmp):
    import atomai_amd as aoi
    X = _images()[0][:, 0].astype(np.float32)
    cls = aoi.models.VAE if kind == "VAE" else aoi.models.rVAE
    m = cls((8, 8), numhidden_encoder=16, numhidden_decoder=16)
    m.fit(X, training_cycles=4, batch_size=2, filename=str(tmp / "vae_metadict"))
    loaded = aoi.models.load_model(str(tmp / "vae_metadict.tar"))
    for a, b in ((loaded.encoder_net, m.encoder_net), (loaded.decoder_net, m.decoder_net)):
        for p1, p2 in zip(a.parameters(), b.parameters()):
            assert np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy())
    assert _opt_equal(m.optim, loaded.optim)
    loss0 = abs(m.loss_history["train_loss"][0])             # resume: the loaded model keeps improving
    loaded.fit(X, training_cycles=4, batch_size=2, filename=str(tmp / "vae_metadict"))
    loss1 = abs(loaded.loss_history["train_loss"][0])
    assert not np.isnan(loss1) and loss1 < loss0


# ------------------------------------------------------------------ trainers/test_etrainer.py, predictors/test_epredictor.py
def ensemble_seg(model, binary, full_epoch, tmp, **kw):
    import atomai_amd as aoi
    X, Xt = _images()
    y, yt = _labels(binary)
    et = aoi.trainers.EnsembleTrainer(model, nb_classes=1 if binary else 3, upsampling="nearest", **kw)
    et.compile_ensemble_trainer(training_cycles=4, full_epoch=full_epoch, batch_size=2, filename=str(tmp / "model"),
                                plot_training_history=False)
    smodel, ensemble = et.train_ensemble_from_scratch(X, y, Xt, yt, n_models=3)
    for i in ensemble:
        for j in ensemble:
            same = all(np.array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy())
                       for a, b in zip(ensemble[i].values(), ensemble[j].values()))
            assert same == (i == j)
    if not binary and not full_epoch:                         # test_io_ensemble_seg
        _, loaded = aoi.models.load_ensemble(str(tmp / "model_ensemble_metadict.tar"))
        for i in ensemble:
            for a, b in zip(ensemble[i].values(), loaded[i].values()):
                assert np.array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy())


def epredictor_seg(model, tmp, **kw):
    import atomai_amd as aoi
    rs = np.random.RandomState(9)
    X, Xt = rs.random_sample((5, 1, 32, 32)), rs.random_sample((5, 1, 32, 32))
    y, yt = rs.randint(0, 3, (5, 32, 32)), rs.randint(0, 3, (5, 32, 32))
    et = aoi.trainers.EnsembleTrainer(model, batch_norm=False, nb_classes=3, **kw)
    et.compile_ensemble_trainer(training_cycles=32, batch_size=2, compute_accuracy=False, filename=str(tmp / "model"),
                                plot_training_history=False)
    smodel, ensemble = et.train_swag(X, y, Xt, yt, n_models=7)
    mean, var = aoi.predictors.EnsemblePredictor(smodel, ensemble, nb_classes=3).predict(Xt)
    assert mean.shape == var.shape == (5, 32, 32, 3)
