"""Shared bodies of the VAE / rVAE parity tests (emulator tier on CPU, gpu tier on the MI355X)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4

CASES = {
    "rvae16": dict(cls="rVAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32), fit=dict()),
    "rvae16_cap": dict(cls="rVAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32, translation=False,
                                              skip=True), fit=dict(capacity=[5.0, 100, 2.0])),
    "vae16": dict(cls="VAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32), fit=dict()),
    "rvae16_conv": dict(cls="rVAE", ctor=dict(conv_encoder=True, numhidden_encoder=8, numhidden_decoder=32),
                        fit=dict(), file="vae_conv.npz"),
    # class-conditioned models (one-hot label appended to the content latents: rvae.py:131-138, vae.py:677-680)
    "crvae16": dict(cls="rVAE", ctor=dict(nb_classes=3, numhidden_encoder=32, numhidden_decoder=32), fit=dict(),
                    file="vae_cond.npz"),
    "cvae16": dict(cls="VAE", ctor=dict(nb_classes=3, numhidden_encoder=32, numhidden_decoder=32), fit=dict(),
                   file="vae_cond.npz"),
    # plain VAE with the convolutional decoder (Linear -> ConvBlock -> 1x1 conv)
    "vae12_convdec": dict(cls="VAE", ctor=dict(conv_decoder=True, numhidden_encoder=32, numhidden_decoder=8), fit=dict(),
                          file="vae_cond.npz", in_dim=(12, 12)),
    # 3-channel patches, 4 hidden layers of 64 units in the spatial decoder
    "rvae12_rgb": dict(cls="rVAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=64, numlayers_decoder=4),
                       fit=dict(), file="vae_cond.npz", in_dim=(12, 12, 3)),
    # reconstruction loss 'ce' (vi_losses.py:27-34): per-sample sums for 2-D patches, the reference's channel-sum /
    # pixel-mean quirk for 3-D ones, and the capacity form (per-sample term vectors instead of the fused scalar)
    "rvae16_ce": dict(cls="rVAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32), fit=dict(), loss="ce",
                      file="vae_ce.npz"),
    "vae16_ce": dict(cls="VAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32), fit=dict(), loss="ce",
                     file="vae_ce.npz"),
    "rvae12_rgb_ce": dict(cls="rVAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32), fit=dict(), loss="ce",
                          file="vae_ce.npz", in_dim=(12, 12, 3)),
    "vae16_ce_cap": dict(cls="VAE", ctor=dict(numhidden_encoder=32, numhidden_decoder=32),
                         fit=dict(capacity=[5.0, 100, 2.0]), loss="ce", file="vae_ce.npz"),
}


def check_vae_case(name, device):
    import atomai_amd as aoi
    c = CASES[name]
    g = np.load(os.path.join(GOLD, c.get("file", "vae.npz")))
    in_dim = c.get("in_dim", (16, 16))
    m = getattr(aoi.models, c["cls"])(in_dim, latent_dim=2, seed=0, **c["ctor"])
    for k, v in m.encoder_net.state_dict().items():        # RNG-order initialisation == reference
        assert np.array_equal(v.cpu().numpy(), g[f"{name}|enc|{k}"]), k
    for k, v in m.decoder_net.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), g[f"{name}|dec|{k}"]), k
    x = g[f"{name}|x"]
    eps_all = torch.from_numpy(g[f"{name}|eps"]).to(device)
    if c["cls"] == "rVAE":                                   # what rVAE.fit sets before the loop
        m.dx_prior = 0.1
        m.kdict_["phi_prior"] = 0.1
    if "capacity" in c["fit"]:
        m.kdict_["capacity"] = c["fit"]["capacity"]
    y = g[f"{name}|y"] if f"{name}|y" in g.files else None
    m.loss = c.get("loss", "mse")                            # what fit(loss=...) sets (rvae.py:196, vae.py:729)
    m.compile_trainer((x, y), None, batch_size=x.shape[0])
    yt = None if y is None else torch.from_numpy(y).long().to(device)
    state = {"i": 0}
    m.reparameterize = lambda zm, zs: zm + zs * eps_all[state["i"]][:, :zm.shape[1]]
    xt = torch.from_numpy(x).to(device)
    elbos = []
    for s in range(3):
        state["i"] = s
        m.encoder_net.train(), m.decoder_net.train()
        m.optim.zero_grad()
        elbo = m.forward_compute_elbo(xt) if yt is None else m.forward_compute_elbo(xt, yt)
        (-elbo).backward()
        if s == 0:
            for which, net in (("enc", m.encoder_net), ("dec", m.decoder_net)):
                for k, p in net.named_parameters():
                    ref = g[f"{name}|g{which}|{k}|f64"]
                    sc = max(np.abs(ref).max(), 1e-30)
                    # no BatchNorm / no kinks on this path: gradients meet the 1e-4 target directly
                    assert np.abs(p.grad.cpu().numpy() - ref).max() / sc < REL_TOL, (which, k)
        m.optim.step()
        elbos.append(elbo.item())
    np.testing.assert_allclose(elbos, g[f"{name}|elbo|f64"], rtol=REL_TOL)
    with torch.no_grad():
        zm, zl = m.encoder_net(xt)
    np.testing.assert_allclose(zm.cpu().numpy(), g[f"{name}|zmean|f64"], rtol=2e-3, atol=2e-4)
    # decode API: shapes as in the reference tests (test/models/test_vae.py)
    if y is None:
        dec = m.decode(np.zeros((3, 2), dtype=np.float32))
        assert dec.shape == (3, *in_dim)
    if f"{name}|decode" in g.files:                          # decode (with labels) after the three steps
        zq = np.array([[0.3, -0.2], [1.0, 0.5], [-0.7, 0.1]], dtype=np.float32)
        dec = m.decode(zq) if y is None else m.decode(zq, np.array([2, 0, 1]))
        ref = g[f"{name}|decode"]
        assert dec.shape == ref.shape
        assert np.abs(dec - ref).max() / np.abs(ref).max() < 20 * REL_TOL      # weights differ at the 1e-4 level
    zmean, zsd = m.encode(x)
    assert zmean.shape == (x.shape[0], m.z_dim) and zsd.shape == zmean.shape


def check_rdecoder_shapes(device, hid, nl, skip, hw, B=2):
    """Other decoder widths / depths / channel counts, a pixel count that is not a multiple of the tile, skip
    connections — in BOTH coordinate modes: explicit (B, n, 2) coordinates (the reference's signature) and the
    in-kernel rotation / translation of the shared grid (forward_grid), whose gradient w.r.t. (phi, dx, dy) must equal
    autograd through transform_coordinates (atomai/utils/coords.py:57-83)."""
    from collections import OrderedDict
    from oracle import vae_oracle as vo
    from atomai_amd.nets import rDecoderNet
    torch.manual_seed(0)
    net = rDecoderNet(hw, 2, nl, hid, bool(skip))
    P = OrderedDict((k, v.double().clone().requires_grad_(True)) for k, v in net.state_dict().items())
    net.to(device)
    grid = vo.imcoordgrid(hw[:2])
    phi, dxy = torch.randn(B), torch.randn(B, 1, 2) * 0.1
    z = torch.randn(B, 2)
    gy = None

    def oracle():
        for v in P.values():
            v.grad = None
        ph, dd = phi.detach().double().clone().requires_grad_(True), dxy.detach().double().clone().requires_grad_(True)
        c2 = vo.transform_coordinates(grid.double().expand(B, *grid.shape), ph, dd)
        c2.retain_grad()
        z2 = z.detach().double().clone().requires_grad_(True)
        yr = vo.r_decoder(P, c2, z2, hw, nl, bool(skip))
        return yr, c2, z2, ph, dd

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max())

    # ---- explicit coordinates
    yr, c2, z2, _, _ = oracle()
    coords = c2.detach().float().to(device).contiguous().requires_grad_(True)
    zt = z.detach().clone().to(device).requires_grad_(True)
    y = net(coords, zt)
    assert y.shape == yr.shape and rel(y, yr) < REL_TOL
    gy = torch.randn(*y.shape)
    y.backward(gy.to(device))
    yr.backward(gy.double())
    for a, b in [(coords.grad, c2.grad), (zt.grad, z2.grad)] + [(p.grad, P[k].grad) for k, p in net.named_parameters()]:
        assert rel(a, b) < REL_TOL
    # ---- rotation / translation inside the kernel
    net.zero_grad()
    yr, c2, z2, ph, dd = oracle()
    theta = torch.cat((phi[:, None], dxy[:, 0]), 1).detach().clone().to(device).requires_grad_(True)
    zt = z.detach().clone().to(device).requires_grad_(True)
    y = net.forward_grid(grid.to(device), theta, zt)
    assert rel(y, yr) < REL_TOL
    y.backward(gy.to(device))
    yr.backward(gy.double())
    ref_theta = torch.cat((ph.grad[:, None], dd.grad[:, 0]), 1)
    assert rel(theta.grad, ref_theta) < REL_TOL
    for a, b in [(zt.grad, z2.grad)] + [(p.grad, P[k].grad) for k, p in net.named_parameters()]:
        assert rel(a, b) < REL_TOL


def check_rdecoder_saved_equals_recompute(device, hid, nl, skip, hw, B=3):
    """The two backward variants of the spatial decoder — hidden activations kept by the forward kernel and read back
    (default in training) vs recomputed per tile (AMX_RDEC_SAVE=0) — must give BIT-identical outputs and gradients."""
    import atomai_amd.nets.ed as ed
    from oracle import vae_oracle as vo
    from atomai_amd.nets import rDecoderNet
    torch.manual_seed(1)
    net = rDecoderNet(hw, 2, nl, hid, bool(skip)).to(device)
    grid = vo.imcoordgrid(hw[:2]).to(device)
    theta0 = torch.cat((torch.randn(B, 1), 0.1 * torch.randn(B, 2)), 1)
    z0 = torch.randn(B, 2)
    res = []
    prev = ed.RDEC_SAVE[0]
    try:
        for save in (True, False):
            ed.RDEC_SAVE[0] = save
            net.zero_grad()
            theta = theta0.clone().to(device).requires_grad_(True)
            zt = z0.clone().to(device).requires_grad_(True)
            y = net.forward_grid(grid, theta, zt)
            if save and len(res) == 0:
                gy = torch.randn(*y.shape)
            y.backward(gy.to(device))
            res.append([y.detach().cpu(), theta.grad.cpu(), zt.grad.cpu()] + [p.grad.cpu().clone() for p in net.parameters()])
    finally:
        ed.RDEC_SAVE[0] = prev
    for a, b in zip(*res):
        assert torch.equal(a, b)


def check_rvae_fused_latent_path(device):
    """rVAE.forward_compute_elbo with the default reparameterize takes the fused path (one kernel for z = mean + sd * eps,
    the (phi, dx, dy) / content split and the translation prior; scalar ELBO in one combine): same ELBO and gradients as
    the step-by-step torch path fed with the same eps, with and without translation."""
    import atomai_amd as aoi
    for translation in (True, False):
        m = aoi.models.rVAE((16, 16), latent_dim=2, seed=0, translation=translation, numhidden_encoder=32,
                            numhidden_decoder=32)
        m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
        assert m._default_reparameterize()
        x = torch.rand(6, 16, 16).to(device)
        params = list(m.encoder_net.parameters()) + list(m.decoder_net.parameters())
        res = []
        for fused in (True, False):
            torch.manual_seed(3)
            for p in params:
                p.grad = None
            if not fused:                                        # an instance-level override selects the torch path
                m.reparameterize = lambda zm, zs: zm + zs * zm.new(zm.size(0), zm.size(1)).normal_()
                assert not m._default_reparameterize()
            m.encoder_net.train(), m.decoder_net.train()
            elbo = m.forward_compute_elbo(x)
            (-elbo).backward()
            res.append((elbo.item(), [p.grad.detach().cpu().clone() for p in params]))
        assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[1][0])
        for a, b in zip(res[0][1], res[1][1]):
            assert float((a - b).abs().max()) <= 1e-5 * max(1e-6, float(b.abs().max()))


def check_rd_tanh_bound(device, bound=2e-7):
    """Element-wise ABSOLUTE error of the rDecoder kernels' activation function (csrc/rdecoder.hip `rd_tanh`: hardware
    exp + reciprocal, 1 - 2 / (e^2x + 1)) against float64 tanh over [-20, 20] — a dense grid, a log grid down to the
    smallest normal numbers on both sides of 0, subnormal inputs, +-0 and the saturation range — through
    amx_rdec_tanh_probe.  The header comment of rd_tanh claims <= 2e-7 (no test stated it)."""
    from atomai_amd import _lib as L
    tiny = np.float32(np.finfo(np.float32).tiny)
    xs = np.concatenate([
        np.linspace(-20, 20, 2_000_001, dtype=np.float64).astype(np.float32),
        np.float32(10.0) ** np.linspace(-37.9, 1.3, 200_001, dtype=np.float64).astype(np.float32),
        -(np.float32(10.0) ** np.linspace(-37.9, 1.3, 200_001, dtype=np.float64).astype(np.float32)),
        np.array([0.0, -0.0, tiny, -tiny, tiny / 2, -tiny / 2, tiny * 1.5, 1e-45, -1e-45, 8.5, 9.0, 9.5, 10.0, 15.0, 20.0,
                  -8.5, -9.0, -15.0, -20.0, 44.0, -44.0, 88.0, -88.0, 100.0, -100.0], dtype=np.float32)])
    x = torch.from_numpy(xs).to(device)
    y = torch.empty_like(x)
    L.call("amx_rdec_tanh_probe", L.ptr(x), L.ptr(y), x.numel(), L.stream_ptr(x))
    ref = np.tanh(xs.astype(np.float64))
    got = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    i = int(err.argmax())
    assert err[i] <= bound, (xs[i], got[i], ref[i], err[i])
    assert (np.abs(got) <= 1.0).all()                        # saturates to +-1, never beyond
    return float(err[i]), float(xs[i])
