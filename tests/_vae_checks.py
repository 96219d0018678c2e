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
}


def check_vae_case(name, device):
    import atomai_amd as aoi
    c = CASES[name]
    g = np.load(os.path.join(GOLD, c.get("file", "vae.npz")))
    m = getattr(aoi.models, c["cls"])((16, 16), latent_dim=2, seed=0, **c["ctor"])
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
    m.compile_trainer((x, None), None, batch_size=x.shape[0])
    state = {"i": 0}
    m.reparameterize = lambda zm, zs: zm + zs * eps_all[state["i"]][:, :zm.shape[1]]
    xt = torch.from_numpy(x).to(device)
    elbos = []
    for s in range(3):
        state["i"] = s
        m.encoder_net.train(), m.decoder_net.train()
        m.optim.zero_grad()
        elbo = m.forward_compute_elbo(xt)
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
    dec = m.decode(np.zeros((3, 2), dtype=np.float32))
    assert dec.shape == (3, 16, 16)
    zmean, zsd = m.encode(x)
    assert zmean.shape == (x.shape[0], m.z_dim) and zsd.shape == zmean.shape
