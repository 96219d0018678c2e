"""Pins oracle/vae_oracle.py against golden vectors generated from the real reference (tests/golden/vae.npz)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo

CASES = {
    "rvae16": dict(kind="rvae", translation=True, skip=False, capacity=None),
    "rvae16_cap": dict(kind="rvae", translation=False, skip=True, capacity=[5.0, 100, 2.0]),
    "vae16": dict(kind="vae", capacity=None),
    "rvae16_conv": dict(kind="rvae", translation=True, skip=False, capacity=None, conv=True, file="vae_conv.npz"),
    # reconstruction loss 'ce' (vi_losses.py:27-34; the reference needs np.product restored under numpy 2, ref_harness.py)
    "rvae16_ce": dict(kind="rvae", translation=True, skip=False, capacity=None, loss="ce", file="vae_ce.npz"),
    "vae16_ce": dict(kind="vae", capacity=None, loss="ce", file="vae_ce.npz"),
    "rvae12_rgb_ce": dict(kind="rvae", translation=True, skip=False, capacity=None, loss="ce", file="vae_ce.npz",
                          grid=(12, 12)),
    "vae16_ce_cap": dict(kind="vae", capacity=[5.0, 100, 2.0], loss="ce", file="vae_ce.npz"),
}


def _sd(g, prefix, dtype):
    return OrderedDict((k[len(prefix):], torch.from_numpy(g[k]).to(dtype)) for k in g.files if k.startswith(prefix))


def test_coordinate_helpers(golden_dir):
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    assert np.array_equal(vo.imcoordgrid((7, 5)).numpy(), g["grid_7x5"])
    assert np.array_equal(vo.imcoordgrid((16, 16)).numpy(), g["grid_16x16"])
    out = vo.transform_coordinates(vo.imcoordgrid((7, 5)).expand(3, 35, 2), torch.from_numpy(g["tc_phi"]),
                                   torch.from_numpy(g["tc_dx"]))
    np.testing.assert_allclose(out.numpy(), g["tc_out"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_elbo_and_grads(golden_dir, name, tag, dtype, tol):
    c = CASES[name]
    g = np.load(os.path.join(golden_dir, c.get("file", "vae.npz")))
    enc, dec = _sd(g, f"{name}|enc|", dtype), _sd(g, f"{name}|dec|", dtype)
    x = torch.from_numpy(g[f"{name}|x"]).to(dtype)
    eps = torch.from_numpy(g[f"{name}|eps"][0]).to(dtype)
    leaves = {("e", k): v.clone().requires_grad_(True) for k, v in enc.items()}
    leaves.update({("d", k): v.clone().requires_grad_(True) for k, v in dec.items()})
    e = {k: leaves[("e", k)] for k in enc}
    d = {k: leaves[("d", k)] for k in dec}
    if c["kind"] == "rvae":
        elbo = vo.rvae_forward_elbo(e, d, x, eps, vo.imcoordgrid(c.get("grid", (16, 16)), dtype), c["translation"], 0.1,
                                    0.1, c["skip"], c["capacity"], num_iter=1, conv_enc=c.get("conv", False),
                                    loss=c.get("loss", "mse"))
    else:
        elbo = vo.vae_forward_elbo(e, d, x, eps, c["capacity"], num_iter=1, loss=c.get("loss", "mse"))
    np.testing.assert_allclose(float(elbo), g[f"{name}|elbo|{tag}"][0], rtol=tol)
    (-elbo).backward()
    for (which, k), v in leaves.items():
        ref = g[f"{name}|g{'enc' if which == 'e' else 'dec'}|{k}|{tag}"]
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(v.grad.numpy() - ref).max() / scale < 50 * tol, (which, k)
