"""Generates the golden fixtures under tests/golden/ by importing the REAL reference
(pycroscopy/atomai at /root/reference) in the dev container.  Run once here:

    python oracle/make_golden.py [seg] [blocks] [config1] [predict] [vae] [gp] ...

The reference's Python never travels to the GPU box; only these small .npz vectors do.
Every network fixture is emitted twice: reference in fp32 and the same module ``.double()``'d,
so each golden carries its own fp32 noise floor (SURVEY.md §7 "Parity budget").
TEST INFRASTRUCTURE ONLY.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def _np(d, suffix=""):
    return {k + suffix: v.detach().cpu().numpy().copy() for k, v in d.items()}


def _seg_case(aoi, name, model, nb_classes, nb_filters, B, H, seed, upsampling="bilinear",
              with_dilation=False, steps=3, layers=None):
    from atomai.nets import init_fcnn_model
    from atomai.losses_metrics import select_loss
    from atomai.utils import set_train_rng
    out = {}
    rs = np.random.RandomState(seed + 100)
    x = rs.rand(B, 1, H, H).astype(np.float32)
    if nb_classes == 1:
        y = (rs.rand(B, 1, H, H) > 0.5).astype(np.float32)
    else:
        y = rs.randint(0, nb_classes, (B, H, H)).astype(np.int64)
    out["x"], out["y"] = x, y
    kw = dict(nb_filters=nb_filters, upsampling=upsampling)
    if model == "Unet":
        kw["with_dilation"] = with_dilation
    if layers is not None:
        kw["layers"] = layers
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        set_train_rng(seed)                                   # trainer.py:659-661
        net, meta = init_fcnn_model(model, nb_classes, **kw)
        if tag == "f32":
            out.update(_np(net.state_dict(), "|init"))
        net = net.to(dt)
        crit = select_loss("ce", nb_classes)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        xt = torch.from_numpy(x).to(dt)
        yt = torch.from_numpy(y) if nb_classes > 1 else torch.from_numpy(y).to(dt)
        losses = []
        for s in range(steps):
            net.train()
            opt.zero_grad()
            logits = net(xt)
            loss = crit(logits, yt)
            loss.backward()
            if s == 0:
                out["logits|" + tag] = logits.detach().numpy()
                out.update({k + "|grad|" + tag: p.grad.detach().numpy().copy()
                            for k, p in net.named_parameters()})
            opt.step()
            losses.append(loss.item())
            if s == 0:
                out.update({k + "|bn1|" + tag: v.detach().numpy().copy()
                            for k, v in net.state_dict().items() if "running" in k})
        out["losses|" + tag] = np.array(losses)
        out.update(_np(net.state_dict(), "|after|" + tag))
        net.eval()
        with torch.no_grad():
            out["eval_logits|" + tag] = net(xt).numpy()
    out["meta"] = np.array([nb_classes, nb_filters, B, H, seed, int(with_dilation)])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "losses f32", out["losses|f32"], "f64", out["losses|f64"])


def make_seg_res(aoi):
    """SegResNet: residual blocks with BatchNorm BEFORE the activation."""
    _seg_case(aoi, "seg_segresnet_c3_nf4_b2_32", "SegResNet", 3, 4, 2, 32, 1)
    _seg_case(aoi, "seg_segresnet_c1_nf4_b2_16_nearest", "SegResNet", 1, 4, 2, 16, 3, upsampling="nearest")


def make_seg_hed(aoi):
    """ResHedNet: three residual stages, BatchNorm'ed 1x1 side outputs interpolated to the input size (x2 / x4)."""
    _seg_case(aoi, "seg_reshednet_c3_nf4_b2_32", "ResHedNet", 3, 4, 2, 32, 1, layers=[2, 2, 2])
    _seg_case(aoi, "seg_reshednet_c3_nf4_b2_22", "ResHedNet", 3, 4, 2, 22, 5, layers=[1, 1, 1])
    _seg_case(aoi, "seg_reshednet_c1_nf4_b2_22_nearest", "ResHedNet", 1, 4, 2, 22, 4, upsampling="nearest",
              layers=[1, 2, 1])


def make_seg(aoi):
    _seg_case(aoi, "seg_unet_c3_nf4_b2_32", "Unet", 3, 4, 2, 32, 1)
    _seg_case(aoi, "seg_unet_c1_nf4_b2_16_nearest", "Unet", 1, 4, 2, 16, 2, upsampling="nearest")
    _seg_case(aoi, "seg_unet_dil_c3_nf4_b2_32", "Unet", 3, 4, 2, 32, 1, with_dilation=True)
    _seg_case(aoi, "seg_dilnet_c1_nf5_b2_32", "dilnet", 1, 5, 2, 32, 1)
    make_seg_res(aoi)
    make_seg_hed(aoi)
    # default-width nets: pin the RNG-order initialisation by per-tensor moments only
    from atomai.nets import init_fcnn_model
    from atomai.utils import set_train_rng
    out = {}
    for model, ncls in (("Unet", 3), ("dilnet", 1)):
        set_train_rng(1)
        net, meta = init_fcnn_model(model, ncls)
        for k, v in net.state_dict().items():
            v = v.double()
            out[f"{model}|{k}"] = np.array([v.numel(), v.sum().item(), (v * v).sum().item(),
                                            v.flatten()[0].item(), v.flatten()[-1].item()])
        out[f"{model}|nparams"] = np.array(sum(p.numel() for p in net.parameters()))
    np.savez_compressed(os.path.join(GOLD, "seg_default_init_moments.npz"), **out)


def make_blocks(aoi):
    from atomai.nets.blocks import ConvBlock, UpsampleBlock, DilatedBlock
    out = {}
    torch.manual_seed(7)
    rs = np.random.RandomState(7)
    cases = {
        "convblock_bn": (lambda: ConvBlock(2, 2, 6, 8, batch_norm=True), (2, 6, 12, 12)),
        "convblock_nobn_a01": (lambda: ConvBlock(2, 2, 1, 8, lrelu_a=0.1), (2, 1, 12, 12)),
        "up_bilinear": (lambda: UpsampleBlock(2, 8, 4, mode="bilinear"), (2, 8, 6, 10)),
        "up_nearest": (lambda: UpsampleBlock(2, 8, 4, mode="nearest"), (2, 8, 6, 10)),
        "dilated_bn": (lambda: DilatedBlock(2, 6, 8, [2, 4, 6], [2, 4, 6], batch_norm=True),
                       (2, 6, 16, 16)),
    }
    for name, (ctor, shp) in cases.items():
        m = ctor()
        x = rs.randn(*shp).astype(np.float32)
        gy = None
        out[f"{name}|x"] = x
        for k, v in m.state_dict().items():
            out[f"{name}|sd|{k}"] = v.numpy().copy()
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            mm = copy.deepcopy(m).to(dt)
            for mode in ("train", "eval"):
                mm.train(mode == "train")
                xt = torch.from_numpy(x).to(dt).requires_grad_(True)
                y = mm(xt)
                if gy is None:
                    gy = rs.randn(*y.shape).astype(np.float32)
                    out[f"{name}|gy"] = gy
                mm.zero_grad()
                y.backward(torch.from_numpy(gy).to(dt))
                out[f"{name}|y|{mode}|{tag}"] = y.detach().numpy()
                out[f"{name}|gx|{mode}|{tag}"] = xt.grad.numpy().copy()
                for k, p in mm.named_parameters():
                    out[f"{name}|gp|{k}|{mode}|{tag}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "seg_blocks.npz"), **out)
    print("blocks:", len(out), "arrays")


def make_config1(aoi):
    """BASELINE.json configs[0]: Segmentor U-Net nb_classes=3, 8x(256x256), 10 cycles, CPU."""
    rs = np.random.RandomState(0)
    X = rs.rand(8, 256, 256).astype(np.float32)
    y = rs.randint(0, 3, (8, 256, 256))
    Xt = rs.rand(8, 256, 256).astype(np.float32)
    yt = rs.randint(0, 3, (8, 256, 256))
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        m = aoi.models.Segmentor(nb_classes=3)
        m.fit(X, y, Xt, yt, training_cycles=10, batch_size=8, plot_training_history=False)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "seg_config1_losses.npz"),
                        train_loss=np.array(m.loss_acc["train_loss"]),
                        test_loss=np.array(m.loss_acc["test_loss"]),
                        batch_idx_train=np.array(m.batch_idx_train),
                        batch_idx_test=np.array(m.batch_idx_test))
    print("config1 train", m.loss_acc["train_loss"])


def make_predict(aoi):
    from atomai.utils import img_pad, torch_format_image
    from atomai.nets import init_fcnn_model
    from atomai.utils import set_train_rng
    from atomai.predictors import SegPredictor
    out = {}
    rs = np.random.RandomState(3)
    img = (rs.rand(3, 13, 18) * 7 - 2).astype(np.float32)
    out["img"] = img
    for f in (2, 8):
        out[f"pad{f}"] = img_pad(img.copy(), f)
        out[f"fmt{f}"] = torch_format_image(img_pad(img.copy(), f)).numpy()
    for model, ncls, nf in (("Unet", 3, 4), ("dilnet", 1, 5)):
        set_train_rng(1)
        net, _ = init_fcnn_model(model, ncls, nb_filters=nf)
        # make running stats non-trivial so eval-mode BN is actually exercised
        net.train()
        with torch.no_grad():
            for _ in range(2):
                net(torch.from_numpy(rs.rand(2, 1, 16, 24).astype(np.float32)))
        for k, v in net.state_dict().items():
            out[f"{model}|sd|{k}"] = v.numpy().copy()
        p = SegPredictor(net, use_gpu=False, nb_classes=ncls,
                         downsampling=8 if model == "Unet" else 2, verbose=False)
        out[f"{model}|probs"] = p.run(img, compute_coords=False, num_batches=2)
    np.savez_compressed(os.path.join(GOLD, "seg_predict.npz"), **out)
    print("predict ok", out["Unet|probs"].shape, out["dilnet|probs"].shape)


def _vae_run(out, rs, name, ctor, fit_kw, in_dim, B, steps=3, nb_classes=0, loss="mse"):
    x = rs.rand(B, *in_dim).astype(np.float32)
    eps_all = rs.randn(steps, B, 8).astype(np.float32)
    out[f"{name}|x"], out[f"{name}|eps"] = x, eps_all
    y = None
    if nb_classes:
        y = np.arange(B) % nb_classes
        rs.shuffle(y)
        out[f"{name}|y"] = y
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        m = ctor()
        if tag == "f32":
            for k, v in m.encoder_net.state_dict().items():
                out[f"{name}|enc|{k}"] = v.numpy().copy()
            for k, v in m.decoder_net.state_dict().items():
                out[f"{name}|dec|{k}"] = v.numpy().copy()
        m.encoder_net.to(dt), m.decoder_net.to(dt)
        if hasattr(m, "x_coord"):
            m.x_coord = m.x_coord.to(dt)
        # what rVAE.fit / VAE.fit set up before the epoch loop (rvae.py:191-200, vae.py:722-731)
        if hasattr(m, "translation"):
            m.dx_prior = fit_kw.get("translation_prior", 0.1)
            m.kdict_["phi_prior"] = fit_kw.get("rotation_prior", 0.1)
        if "capacity" in fit_kw:
            m.kdict_["capacity"] = fit_kw["capacity"]
        m.loss = loss                                          # what fit(loss=...) sets (rvae.py:196, vae.py:729)
        m.compile_trainer((x, y), None, batch_size=B)
        yt = None if y is None else torch.from_numpy(y).long()
        state = {"i": 0}

        def reparam(z_mean, z_sd, st=state, e=eps_all, d=dt):
            ee = torch.from_numpy(e[st["i"]][:, :z_mean.shape[1]]).to(d)
            return z_mean + z_sd * ee
        m.reparameterize = reparam
        xt = torch.from_numpy(x).to(dt)
        elbos = []
        for s in range(steps):
            state["i"] = s
            m.encoder_net.train(), m.decoder_net.train()
            m.optim.zero_grad()
            elbo = m.forward_compute_elbo(xt) if yt is None else m.forward_compute_elbo(xt, yt)
            (-elbo).backward()
            if s == 0:
                for k, p in m.encoder_net.named_parameters():
                    out[f"{name}|genc|{k}|{tag}"] = p.grad.numpy().copy()
                for k, p in m.decoder_net.named_parameters():
                    out[f"{name}|gdec|{k}|{tag}"] = p.grad.numpy().copy()
            m.optim.step()
            elbos.append(elbo.item())
        out[f"{name}|elbo|{tag}"] = np.array(elbos)
        for k, v in m.encoder_net.state_dict().items():
            out[f"{name}|enc_after|{k}|{tag}"] = v.numpy().copy()
        for k, v in m.decoder_net.state_dict().items():
            out[f"{name}|dec_after|{k}|{tag}"] = v.numpy().copy()
        # encode / decode (eval) with the trained nets
        with torch.no_grad():
            zm, zs = m.encoder_net(xt)
        out[f"{name}|zmean|{tag}"], out[f"{name}|zlogsd|{tag}"] = zm.numpy(), zs.numpy()
        if tag == "f64":                                      # decode API on the trained generative model
            m.encoder_net.float(), m.decoder_net.float()
            if hasattr(m, "x_coord"):
                m.x_coord = m.x_coord.float()
            zq = np.array([[0.3, -0.2], [1.0, 0.5], [-0.7, 0.1]], dtype=np.float32)
            out[f"{name}|decode"] = m.decode(zq) if not nb_classes else m.decode(zq, np.array([2, 0, 1]))
    print(name, "elbo f32", out[f"{name}|elbo|f32"], "f64", out[f"{name}|elbo|f64"])



def make_vae(aoi):
    from atomai.utils import set_train_rng
    from atomai.utils.coords import imcoordgrid, transform_coordinates
    from atomai.losses_metrics import rvae_loss, vae_loss
    out = {}
    rs = np.random.RandomState(5)
    out["grid_7x5"] = imcoordgrid((7, 5)).numpy()
    out["grid_16x16"] = imcoordgrid((16, 16)).numpy()
    phi = torch.from_numpy(rs.randn(3).astype(np.float32))
    dx = torch.from_numpy(rs.randn(3, 1, 2).astype(np.float32) * 0.1)
    g = imcoordgrid((7, 5)).expand(3, 35, 2)
    out["tc_phi"], out["tc_dx"] = phi.numpy(), dx.numpy()
    out["tc_out"] = transform_coordinates(g, phi, dx).numpy()

    def run(*a, **k):
        _vae_run(out, rs, *a, **k)

    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        run("rvae16", lambda: aoi.models.rVAE((16, 16), latent_dim=2, seed=0,
                                              numhidden_encoder=32, numhidden_decoder=32),
            dict(), (16, 16), 6)
        run("rvae16_cap", lambda: aoi.models.rVAE((16, 16), latent_dim=2, seed=0,
                                                  numhidden_encoder=32, numhidden_decoder=32,
                                                  translation=False, skip=True),
            dict(capacity=[5.0, 100, 2.0]), (16, 16), 4)
        run("vae16", lambda: aoi.models.VAE((16, 16), latent_dim=2, seed=0,
                                            numhidden_encoder=32, numhidden_decoder=32),
            dict(), (16, 16), 6)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "vae.npz"), **out)


def make_vae_ce(aoi):
    """Reconstruction loss 'ce' (vi_losses.py:27-34): binary cross entropy with logits.  For a 2-D in_dim it is summed
    over the pixels of a sample; for a 3-D in_dim the reference's reshape makes it a sum over the CHANNELS only, so the
    later .mean() runs over samples x pixels — both pinned here."""
    out = {}
    rs = np.random.RandomState(11)

    def run(*a, **k):
        _vae_run(out, rs, *a, **k)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        run("rvae16_ce", lambda: aoi.models.rVAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=32,
                                                 numhidden_decoder=32), dict(), (16, 16), 6, loss="ce")
        run("vae16_ce", lambda: aoi.models.VAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=32,
                                               numhidden_decoder=32), dict(), (16, 16), 6, loss="ce")
        run("rvae12_rgb_ce", lambda: aoi.models.rVAE((12, 12, 3), latent_dim=2, seed=0, numhidden_encoder=32,
                                                     numhidden_decoder=32), dict(), (12, 12, 3), 5, loss="ce")
        run("vae16_ce_cap", lambda: aoi.models.VAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=32,
                                                   numhidden_decoder=32), dict(capacity=[5.0, 100, 2.0]), (16, 16), 4,
            loss="ce")
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "vae_ce.npz"), **out)


def make_seg_k9(aoi):
    """nb_classes = 9 (> the 8 classes the register-resident head kernels hold): px conv, CrossEntropyLoss, one step."""
    _seg_case(aoi, "seg_unet_c9_nf4_b2_32", "Unet", 9, 4, 2, 32, 1)
    _seg_case(aoi, "seg_dilnet_c11_nf5_b2_32", "dilnet", 11, 5, 2, 32, 2)


def make_vae_cond(aoi):
    """Class-conditioned (r)VAE (rvae.py:131-138, vae.py:677-680), a multi-channel / 4-layer spatial decoder."""
    out = {}
    rs = np.random.RandomState(11)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        _vae_run(out, rs, "crvae16", lambda: aoi.models.rVAE((16, 16), latent_dim=2, nb_classes=3, seed=0,
                                                             numhidden_encoder=32, numhidden_decoder=32),
                 dict(), (16, 16), 6, nb_classes=3)
        _vae_run(out, rs, "cvae16", lambda: aoi.models.VAE((16, 16), latent_dim=2, nb_classes=3, seed=0,
                                                           numhidden_encoder=32, numhidden_decoder=32),
                 dict(), (16, 16), 6, nb_classes=3)
        _vae_run(out, rs, "vae12_convdec", lambda: aoi.models.VAE((12, 12), latent_dim=2, seed=0, conv_decoder=True,
                                                                  numhidden_encoder=32, numhidden_decoder=8),
                 dict(), (12, 12), 4)
        _vae_run(out, rs, "rvae12_rgb", lambda: aoi.models.rVAE((12, 12, 3), latent_dim=2, seed=0,
                                                                numhidden_encoder=32, numhidden_decoder=64,
                                                                numlayers_decoder=4),
                 dict(), (12, 12, 3), 4)
    finally:
        os.chdir(cwd)
    out = {k: v for k, v in out.items() if "_after|" not in k}          # not asserted on: keep the fixture small
    np.savez_compressed(os.path.join(GOLD, "vae_cond.npz"), **out)


def make_augment(aoi):
    """datatransform.run of the real reference for the steps that are pure numpy / scipy there (poisson, blur,
    background, the two min-max normalisations); per-pixel poisson draws are recorded by wrapping np.random.poisson."""
    from atomai.transforms import datatransform
    out = {}
    rs = np.random.RandomState(3)
    # (1) blur + background on a batch: every scalar draw is reproducible from the seed
    X = rs.rand(3, 20, 28)
    y = (rs.rand(3, 20, 28, 1) > 0.5).astype(np.float64)
    dt = datatransform(1, "channel_last", "channel_first", False, 7, blur=[1, 50], background=True)
    Xa, ya = dt.run(X.copy(), y.copy())
    out["bb|X"], out["bb|out"], out["bb|seed"] = X, Xa, np.array(7)
    # (2) poisson on ONE image (the per-pixel draws consume the global stream, so only the first image's level is
    # reproducible from the seed); the draws themselves are captured
    X1 = rs.rand(1, 24, 24)
    rec = []
    orig = np.random.poisson

    def wrapped(lam, *a, **k):
        r = orig(lam, *a, **k)
        rec.append(np.array(r))
        return r
    np.random.poisson = wrapped
    try:
        dt = datatransform(1, "channel_last", "channel_first", False, 11, poisson_noise=[30, 40])
        Xp, _ = dt.run(X1.copy(), y[:1, :24, :24].copy())
    finally:
        np.random.poisson = orig
    out["po|X"], out["po|out"], out["po|draws"], out["po|seed"] = X1, Xp, rec[0].astype(np.float64)[None], np.array(11)
    # (3) jitter (+ background): rows rolled by scipy.stats.poisson.rvs shifts, which draw from numpy's global stream
    # and are therefore reproducible from the seed; recorded as well
    from scipy import stats
    Xj = rs.rand(3, 20, 28)
    shifts = []
    orig_rvs = stats.poisson.rvs

    def rvs(*a, **k):
        r = orig_rvs(*a, **k)
        shifts.append(np.array(r))
        return r
    stats.poisson.rvs = rvs
    try:
        dt = datatransform(1, "channel_last", "channel_first", False, 13, jitter=[20, 50], background=True)
        Xjo, _ = dt.run(Xj.copy(), y.copy())
    finally:
        del stats.poisson.rvs
    out["ji|X"], out["ji|out"], out["ji|shifts"], out["ji|seed"] = Xj, Xjo, np.stack(shifts).astype(np.int64), np.array(13)
    np.savez_compressed(os.path.join(GOLD, "augment.npz"), **out)
    print("augment ok", Xa.shape, Xp.shape, Xjo.shape, np.stack(shifts).max())


def make_vae_conv(aoi):
    """rVAE with the opt-in convolutional encoder (ed.py:231-289): ConvBlock(lrelu 0.1, no BN) + two Linear."""
    out = {}
    rs = np.random.RandomState(11)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        _vae_run(out, rs, "rvae16_conv",
                 lambda: aoi.models.rVAE((16, 16), latent_dim=2, seed=0, conv_encoder=True,
                                         numhidden_encoder=8, numhidden_decoder=32),
                 dict(), (16, 16), 4)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "vae_conv.npz"), **out)


def make_ckpt(aoi):
    """Checkpoints written by the reference's own save_model (trainer.py:344-358, vitrainer.py:361-377) +
    what the reference predicts / encodes after load_model (models/loaders.py:25-195)."""
    import shutil
    out = {}
    rs = np.random.RandomState(21)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        X, Xt = rs.rand(6, 16, 16).astype(np.float32), rs.rand(4, 16, 16).astype(np.float32)
        y, yt = rs.randint(0, 3, (6, 16, 16)), rs.randint(0, 3, (4, 16, 16))
        m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4)
        m.fit(X, y, Xt, yt, training_cycles=3, batch_size=2, filename="refseg", plot_training_history=False)
        shutil.copy("refseg_metadict_final.tar", os.path.join(GOLD, "ref_seg_unet_ckpt.tar"))
        lm = aoi.models.load_model("refseg_metadict_final.tar")
        Xp = rs.rand(2, 16, 16).astype(np.float32)
        out["seg|x"] = Xp
        out["seg|pred"] = lm.predict(Xp, compute_coords=False)
        out["seg|meta_keys"] = np.array(sorted(torch.load("refseg_metadict_final.tar", weights_only=False).keys()))
        v = aoi.models.rVAE((16, 16), latent_dim=2, seed=0, numhidden_encoder=16, numhidden_decoder=16)
        Xv = rs.rand(8, 16, 16).astype(np.float32)
        v.fit(Xv, training_cycles=2, batch_size=4, filename="refrvae")
        shutil.copy("refrvae.tar", os.path.join(GOLD, "ref_rvae_ckpt.tar"))
        lv = aoi.models.load_model("refrvae.tar")
        out["vae|x"] = Xv
        zm, zs = lv.encode(Xv)
        out["vae|zmean"], out["vae|zsd"] = zm, zs
        out["vae|dec"] = lv.decode(np.array([[0.3, -0.2], [1.0, 0.5]], dtype=np.float32))
        out["vae|meta_keys"] = np.array(sorted(torch.load("refrvae.tar", weights_only=False).keys()))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "ckpt.npz"), **out)


def make_locator(aoi):
    """Locator.run of the reference (predictor.py:584-611) on synthetic class maps.  cv2 is not installed in
    this image: `cv_thresh` (utils/img.py:554-564, a one-line cv2.threshold wrapper) is replaced by the
    documented THRESH_BINARY semantics; preprocess / find_com (scipy.ndimage) / rem_edge_coord / dictionary
    assembly are the reference's own code."""
    import atomai.predictors.predictor as rp
    import locator_oracle as lo
    rp.cv_thresh = lambda img, thr=.5: np.where(img > thr, 1, 0).astype(img.dtype)
    rs = np.random.RandomState(33)
    out = {}
    cases = {"c3_48x64": (3, 48, 64, 3, 0.5, 5), "c1_40x40": (2, 40, 40, 1, 0.5, 3),
             "c2_33x47_t07": (2, 33, 47, 2, 0.7, 0), "c3_64x64_dense": (2, 64, 64, 3, 0.3, 8)}
    for name, (B, H, W, C, thr, de) in cases.items():
        x = lo.synthetic_maps(rs, B, H, W, C, n_blobs=40 if "dense" in name else 12)
        if name == "c1_40x40":
            x[1] = 0.0                                   # a frame without any blob
        d = rp.Locator(thr, de).run(x)
        out[f"{name}|x"] = x
        out[f"{name}|cfg"] = np.array([thr, de], dtype=np.float64)
        for i, v in d.items():
            out[f"{name}|coords|{i}"] = v
        print(name, [v.shape for v in d.values()])
    xcf = np.ascontiguousarray(np.transpose(out["c3_48x64|x"], (0, 3, 1, 2)))
    dcf = rp.Locator(0.5, 5, dim_order="channel_first").run(xcf)
    for i, v in dcf.items():
        assert np.array_equal(v, out[f"c3_48x64|coords|{i}"])
    np.savez_compressed(os.path.join(GOLD, "locator.npz"), **out)


def make_ensemble(aoi):
    """EnsembleTrainer of the reference (trainers/etrainer.py): both strategies on a tiny U-Net."""
    out = {}
    rs = np.random.RandomState(41)
    X, Xt = rs.rand(6, 16, 16).astype(np.float32), rs.rand(4, 16, 16).astype(np.float32)
    y, yt = rs.randint(0, 3, (6, 16, 16)), rs.randint(0, 3, (4, 16, 16))
    out["X"], out["y"], out["Xt"], out["yt"] = X, y, Xt, yt
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(training_cycles=3, batch_size=2, plot_training_history=False, filename="ens_a")
        net, ens = et.train_ensemble_from_scratch(X, y, Xt, yt, n_models=2)
        for i, sd in ens.items():
            for k, v in sd.items():
                out[f"scratch|{i}|{k}"] = v.cpu().numpy().copy()
        out["scratch|last_train_loss"] = np.array(et.loss_acc["train_loss"])
        ep = aoi.predictors.EnsemblePredictor(net, ens, nb_classes=3, use_gpu=False)
        out["epred|mean"], out["epred|var"] = ep.predict(Xt, num_batches=2)
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(batch_size=2, plot_training_history=False, filename="ens_b")
        net, ens = et.train_ensemble_from_baseline(X, y, Xt, yt, n_models=2, training_cycles_base=3,
                                                   training_cycles_ensemble=2)
        for i, sd in ens.items():
            for k, v in sd.items():
                out[f"baseline|{i}|{k}"] = v.cpu().numpy().copy()
        for k, v in net.state_dict().items():
            out[f"baseline|avg|{k}"] = v.cpu().numpy().copy()
        out["ens_meta_keys"] = np.array(sorted(torch.load("ens_b_ensemble_metadict.tar", weights_only=False).keys()))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "ensemble.npz"), **out)
    print("ensemble ok", out["scratch|last_train_loss"], out["ens_meta_keys"])


def make_vae_api(aoi):
    """extract_subimages / get_coord_grid / crop_borders (utils/img.py) and BaseVAE.encode_images / reconstruct."""
    from atomai.utils import extract_subimages, get_coord_grid, crop_borders
    out = {}
    rs = np.random.RandomState(51)
    img = rs.rand(2, 20, 24, 2).astype(np.float32)
    coords = {0: np.array([[3.2, 4.7, 0], [0.4, 1.0, 0], [10.5, 12.49, 1], [18.9, 22.0, 0], [9.0, 9.0, 0]]),
              1: np.array([[8.0, 8.0, 0], [15.6, 3.3, 0]])}
    out["img"] = img
    for k, v in coords.items():
        out[f"coords|{k}"] = v
    for ws in (5, 6):
        st, com, fr = extract_subimages(img, coords, ws)
        out[f"sub|{ws}|stack"], out[f"sub|{ws}|com"], out[f"sub|{ws}|frames"] = st, com, fr
    out["grid3"] = get_coord_grid(img[..., 0], 3, return_dict=False)
    out["grid_dict2"] = get_coord_grid(img[0, ..., 0], 7)[0]
    z = -np.ones((9, 11, 2)); z[2:7, 3:9] = rs.rand(5, 6, 2)
    out["crop_in"], out["crop_out"] = z, crop_borders(z, -1)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        v = aoi.models.VAE((8, 8), latent_dim=2, seed=0, numhidden_encoder=16, numhidden_decoder=16)
        X = rs.rand(12, 8, 8).astype(np.float32)
        v.fit(X, training_cycles=2, batch_size=4, filename="vapi")
        for k, t in v.encoder_net.state_dict().items():
            out[f"enc|{k}"] = t.numpy().copy()
        for k, t in v.decoder_net.state_dict().items():
            out[f"dec|{k}"] = t.numpy().copy()
        big = rs.rand(14, 17).astype(np.float32)
        out["big"] = big
        im_, enc_ = v.encode_image_(big, num_batches=3)
        out["encimg|img"], out["encimg|z"] = im_, enc_
        ims, encs = v.encode_images(np.stack([big, big[::-1].copy()]), num_batches=4)
        out["encimgs|img"], out["encimgs|z"] = ims, encs
        torch.manual_seed(3)
        out["recon"] = v.reconstruct(X[:1], num_samples=4)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "vae_api.npz"), **out)
    print("vae_api ok", out["encimg|z"].shape, out["recon"].shape)


def make_dil_drop(aoi):
    """DilatedBlock with Dropout layers (blocks.py:311-312): the block sums the output of EVERY sub-layer, the Dropout
    layer's included (blocks.py:321-329).  The reference's module graph is run unchanged; only the random mask is made
    explicit (torch.nn.functional.dropout -> multiplication by a recorded mask in training, identity in eval)."""
    import torch.nn.functional as F
    from atomai.nets.blocks import DilatedBlock
    out = {}
    rs = np.random.RandomState(21)
    orig = F.dropout
    try:
        for tag, bn, cin, cout, dils, H, W, p in (("bn", True, 3, 8, [2, 4], 14, 12, 0.3),
                                                  ("nobn", False, 5, 6, [2, 4, 6], 13, 17, 0.5)):
            torch.manual_seed(3)
            blk = DilatedBlock(2, cin, cout, dils, dils, batch_norm=bn, dropout_=p)
            out.update({f"{tag}|w|{k}": v.detach().numpy().copy() for k, v in blk.state_dict().items()})
            N = 2
            x = rs.randn(N, cin, H, W).astype(np.float32)
            gy = rs.randn(N, cout, H, W).astype(np.float32)
            masks = [((rs.rand(N, cout, H, W) >= p).astype(np.float32) / (1 - p)) for _ in dils]
            out[f"{tag}|x"], out[f"{tag}|gy"], out[f"{tag}|masks"] = x, gy, np.stack(masks)
            out[f"{tag}|meta"] = np.array([int(bn), cin, cout, H, W, int(round(p * 100))] + dils)
            for dt, dtag in ((torch.float32, "f32"), (torch.float64, "f64")):
                b2 = copy.deepcopy(blk).to(dt).train()
                it = iter(masks)
                F.dropout = lambda inp, p=0.5, training=True, inplace=False: (
                    inp * torch.from_numpy(next(it)).to(inp.dtype) if training else inp)
                xt = torch.from_numpy(x).to(dt).requires_grad_(True)
                y = b2(xt)
                y.backward(torch.from_numpy(gy).to(dt))
                out[f"{tag}|y|{dtag}"] = y.detach().numpy()
                out[f"{tag}|dx|{dtag}"] = xt.grad.numpy()
                out.update({f"{tag}|grad|{k}|{dtag}": q.grad.numpy().copy() for k, q in b2.named_parameters()})
                out.update({f"{tag}|bn|{k}|{dtag}": v.detach().numpy().copy() for k, v in b2.state_dict().items()
                            if "running" in k})
                b2.eval()
                with torch.no_grad():
                    out[f"{tag}|y_eval|{dtag}"] = b2(torch.from_numpy(x).to(dt)).numpy()
            print("dil_drop", tag, out[f"{tag}|y|f64"].shape, float(np.abs(out[f"{tag}|y|f32"] - out[f"{tag}|y|f64"]).max()))
    finally:
        F.dropout = orig
    np.savez_compressed(os.path.join(GOLD, "dilated_dropout.npz"), **out)


def make_iou(aoi):
    """IoU.evaluate of the reference (losses_metrics/metrics.py:16-95) on logits / labels.  cv2 is absent in this image:
    `cv_thresh` (utils/img.py:554-564, a one-line cv2.threshold wrapper) is replaced by the documented THRESH_BINARY
    semantics, as for the Locator golden; threshold_ / squeeze_channels / bincount / Jaccard are the reference's own."""
    import atomai.losses_metrics.metrics as rm
    rm.cv_thresh = lambda img, thr=.5: np.where(img > thr, 1, 0).astype(img.dtype)
    rs = np.random.RandomState(77)
    out = {}
    for name, (N, K, H, W, thr, scale) in {"c3": (3, 3, 20, 24, 0.5, 2.0), "c1": (2, 1, 17, 19, 0.5, 1.5),
                                           "c4_t03": (2, 4, 16, 16, 0.3, 3.0), "c1_t07": (3, 1, 12, 12, 0.7, 2.0),
                                           "c3_missing": (2, 3, 16, 16, 0.5, 2.0)}.items():
        logits = (scale * rs.randn(N, K, H, W)).astype(np.float32)
        if K == 1:
            true = (rs.rand(N, 1, H, W) > 0.5).astype(np.float32)
        else:
            true = rs.randint(0, K if name != "c3_missing" else 2, (N, H, W)).astype(np.int64)
            if name == "c3_missing":
                logits[:, 2] -= 10.0                         # class 2 neither labelled nor predicted: 0 / 1e-10
        out[f"{name}|logits"], out[f"{name}|true"] = logits, true
        out[f"{name}|cfg"] = np.array([K, thr])
        out[f"{name}|iou"] = np.array(rm.IoU(torch.from_numpy(true), torch.from_numpy(logits), True, thr).evaluate())
        print("iou", name, out[f"{name}|iou"])
    np.savez_compressed(os.path.join(GOLD, "iou.npz"), **out)
    # round 5: more than 8 classes, and activation=False (probabilities handed in, metrics.py:37-41 skipped)
    out = {}
    for name, (N, K, H, W, thr, scale, act) in {"c9": (2, 9, 16, 20, 0.5, 4.0, True), "c12_t02": (2, 12, 16, 16, 0.2, 3.0, True),
                                                "c3_noact": (3, 3, 20, 24, 0.5, 2.0, False),
                                                "c1_noact": (2, 1, 17, 19, 0.6, 1.5, False),
                                                "c10_noact": (2, 10, 12, 12, 0.3, 3.0, False)}.items():
        logits = (scale * rs.randn(N, K, H, W)).astype(np.float32)
        pred = logits if act else (torch.softmax(torch.from_numpy(logits), 1).numpy() if K > 1
                                   else torch.sigmoid(torch.from_numpy(logits)).numpy())
        true = ((rs.rand(N, 1, H, W) > 0.5).astype(np.float32) if K == 1
                else rs.randint(0, K, (N, H, W)).astype(np.int64))
        out[f"{name}|pred"], out[f"{name}|true"] = pred, true
        out[f"{name}|cfg"] = np.array([K, thr, int(act)])
        out[f"{name}|iou"] = np.array(rm.IoU(torch.from_numpy(true), torch.from_numpy(pred), act, thr).evaluate())
        print("iou", name, out[f"{name}|iou"])
    np.savez_compressed(os.path.join(GOLD, "iou_wide.npz"), **out)


def make_augment_geom(aoi):
    """rotation -> zoom -> resize of the reference's OWN seg_augmentor / datatransform code (imaug.py:195-227, 253-300,
    302-432).  cv2 is absent in this image: its three entry points are replaced by their documented semantics —
    cv2.flip / cv2.rotate by numpy index reversals / rot90, cv2.resize by oracle/aug_oracle.cv_resize (OpenCV's
    INTER_LINEAR / INTER_CUBIC arithmetic restated; note the shim keeps cv2.resize's real signature, so the reference's
    ``cv2.resize(img, (w, h), rs_method)`` hands rs_method to `dst` and runs INTER_LINEAR, as with the real cv2).  Pinned
    by this golden: draw order and counts under a seed, crop windows, candidate sizes, clipping / rounding, one-hot
    travel of the class maps, squeeze_channels incl. the dropped pairs.  NOT pinned: cv2's arithmetic itself."""
    import atomai.transforms.imaug as ri
    import aug_oracle as ao
    cv2 = ri.cv2
    cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA = 1, 2, 3
    cv2.ROTATE_90_CLOCKWISE, cv2.ROTATE_90_COUNTERCLOCKWISE = 0, 2

    def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=1):
        mode = {1: "linear", 2: "cubic"}[interpolation]
        if src.ndim == 2:
            return ao.cv_resize(src, (dsize[1], dsize[0]), mode)
        out = np.stack([ao.cv_resize(src[..., c], (dsize[1], dsize[0]), mode) for c in range(src.shape[-1])], -1)
        return out[..., 0] if out.shape[-1] == 1 else out       # cv2 drops a trailing singleton channel
    cv2.resize = resize
    cv2.flip = lambda img, code: ao.flip(img, 0 if code == 0 else (1 if code > 0 else -1))
    cv2.rotate = lambda img, code: np.rot90(img, -1 if code == 0 else 1)
    out = {}
    rs = np.random.RandomState(61)
    for name, (K, N, H, W, kw, seed) in {
            "c3_zoom": (3, 6, 48, 48, dict(zoom=True), 3), "c3_resize": (3, 5, 40, 40, dict(resize=True), 4),
            "c3_rot_zoom_resize": (3, 6, 48, 48, dict(rotation=True, zoom=True, resize=[2, 1.5]), 5),
            "c1_zoom_resize": (1, 4, 32, 48, dict(zoom=3, resize=True), 6),
            "c4_all": (4, 5, 64, 64, dict(rotation=True, zoom=True, resize=True), 7),
            "c3_drop": (3, 6, 48, 48, dict(zoom=4, rotation=True), 8)}.items():
        x = rs.rand(N, 1, H, W).astype(np.float32)
        # blob-like class maps so that every class survives most crops
        yy, xx = np.mgrid[0:H, 0:W]
        if K == 1:
            lab = ((np.sin(yy / 5.0)[None] + np.cos(xx / 4.0)[None] + rs.rand(N, 1, 1)) > 0.6).astype(np.float32)[:, None]
        else:
            lab = ((yy[None] // 6 + xx[None] // 5 + rs.randint(0, K, (N, 1, 1))) % K).astype(np.int64)
        if name == "c3_drop":                                # images 1 and 3 hold class 2 only in a corner: lost by most crops
            for i in (1, 3):
                lab[i] = lab[i] % 2
                lab[i, :3, :3] = 2
        aug = ri.seg_augmentor(K, **kw)
        xi, li = aug(torch.from_numpy(x), torch.from_numpy(lab), seed)
        out[f"{name}|x"], out[f"{name}|lab"] = x, lab
        out[f"{name}|cfg"] = np.array([K, seed])
        out[f"{name}|kw"] = np.array(repr(kw))
        out[f"{name}|x_out"], out[f"{name}|lab_out"] = xi.numpy(), li.numpy()
        print("augment_geom", name, tuple(x.shape), "->", tuple(xi.shape), tuple(li.shape))
    np.savez_compressed(os.path.join(GOLD, "augment_geom.npz"), **out)


def make_gp(aoi):
    """fcFeatureExtractor (nets/gp.py:14-26) exactly as dklGPTrainer builds it (gptrainer.py:162-177, 254-262):
    ``set_seed_and_precision`` (utils/nn.py:149-167) seeds numpy / torch with 42 and makes the chosen precision the
    DEFAULT tensor type, so under precision='double' (the trainer's default) the Linear layers are DRAWN in float64.
    Pins state-dict keys, RNG-order init, forward and every gradient.  gpytorch itself is a stub here: nothing of the
    GP layer is (or can be) pinned."""
    from atomai.nets.gp import fcFeatureExtractor
    from atomai.utils import set_seed_and_precision
    out = {}
    rs = np.random.RandomState(5)

    def mom(v):                      # default widths: 5 numbers per tensor instead of 0.6 M values
        v = torch.as_tensor(v).double().flatten()
        return np.array([v.numel(), v.sum().item(), (v * v).sum().item(), v[0].item(), v[-1].item()])
    try:
        for tag, feat, emb, hid, prec, full in (("dbl", 48, 2, None, "double", False),
                                                ("sgl", 48, 2, None, "single", False),
                                                ("small", 20, 3, [32, 16], "double", True),
                                                ("small_sgl", 20, 3, [32, 16], "single", True)):
            set_seed_and_precision(seed=42, precision=prec)
            kw = {} if hid is None else {"hidden_dim": list(hid)}
            net = fcFeatureExtractor(feat, emb, **kw)
            keep = (lambda v: v.detach().numpy().copy()) if full else mom
            out[f"{tag}|keys"] = np.array(list(net.state_dict().keys()))
            out.update({f"{tag}|init|{k}": keep(v) for k, v in net.state_dict().items()})
            x = rs.rand(37, feat)
            w = rs.randn(37, emb)
            out[f"{tag}|x"], out[f"{tag}|w"] = x, w
            out[f"{tag}|meta"] = np.array([feat, emb, int(prec == "double"), int(full)] + (hid or [1000, 500, 50]))
            for dt, dtag in ((torch.float32, "f32"), (torch.float64, "f64")):
                n2 = copy.deepcopy(net).to(dt)
                y = n2(torch.from_numpy(x).to(dt))
                (y * torch.from_numpy(w).to(dt)).sum().backward()
                out[f"{tag}|y|{dtag}"] = y.detach().numpy()
                out.update({f"{tag}|grad|{k}|{dtag}": keep(p.grad) for k, p in n2.named_parameters()})
            print("gp", tag, {k: tuple(v.shape) for k, v in net.state_dict().items()}, net.linear1.weight.dtype)
    finally:
        torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(GOLD, "gp_extractor.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["seg", "blocks", "config1", "predict", "vae", "vae_conv", "ckpt", "locator", "ensemble", "vae_api", "vae_cond", "augment", "gp", "dil_drop", "iou", "augment_geom", "vae_ce", "seg_k9"]
    aoi = ref_harness.import_reference()
    torch.set_num_threads(8)
    for w in what:
        {"seg": make_seg, "blocks": make_blocks, "config1": make_config1,
         "predict": make_predict, "vae": make_vae, "vae_conv": make_vae_conv, "ckpt": make_ckpt, "locator": make_locator, "seg_res": make_seg_res, "seg_hed": make_seg_hed, "ensemble": make_ensemble, "vae_api": make_vae_api, "vae_cond": make_vae_cond, "augment": make_augment, "gp": make_gp, "dil_drop": make_dil_drop, "iou": make_iou, "augment_geom": make_augment_geom, "vae_ce": make_vae_ce, "seg_k9": make_seg_k9}[w](aoi)
    print("done ->", GOLD)
