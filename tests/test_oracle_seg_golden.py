"""Pins oracle/seg_oracle.py against golden vectors generated from the real reference
(oracle/make_golden.py).  CPU only."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import seg_oracle as so

CASES = {
    "seg_unet_c3_nf4_b2_32": dict(model="Unet", kw=dict()),
    "seg_unet_c1_nf4_b2_16_nearest": dict(model="Unet", kw=dict(upsampling="nearest")),
    "seg_unet_dil_c3_nf4_b2_32": dict(model="Unet", kw=dict(with_dilation=True)),
    "seg_dilnet_c1_nf5_b2_32": dict(model="dilnet", kw=dict()),
    "seg_segresnet_c3_nf4_b2_32": dict(model="SegResNet", kw=dict()),
    "seg_segresnet_c1_nf4_b2_16_nearest": dict(model="SegResNet", kw=dict(upsampling="nearest")),
    "seg_reshednet_c3_nf4_b2_32": dict(model="ResHedNet", kw=dict(layers=[2, 2, 2])),
    "seg_reshednet_c3_nf4_b2_22": dict(model="ResHedNet", kw=dict(layers=[1, 1, 1])),
    "seg_reshednet_c1_nf4_b2_22_nearest": dict(model="ResHedNet", kw=dict(upsampling="nearest", layers=[1, 2, 1])),
}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _sd(g, suffix, dtype):
    sd = OrderedDict()
    for k in g.files:
        if k.endswith(suffix):
            v = torch.from_numpy(g[k])
            sd[k[: -len(suffix)]] = v if v.dtype == torch.long else v.to(dtype)
    return sd


@pytest.mark.parametrize("name", list(CASES))
def test_init_matches_reference_rng_order(golden_dir, name):
    g = _load(golden_dir, name)
    ncls, nf, B, H, seed, dil = [int(v) for v in g["meta"]]
    c = CASES[name]
    # set_train_rng(seed) in the reference == torch.manual_seed(seed) for the weight draw
    if c["model"] == "Unet":
        sd = so.init_unet(ncls, nf, with_dilation=bool(dil), seed=seed)
    elif c["model"] == "SegResNet":
        sd = so.init_segresnet(ncls, nf, seed=seed)
    elif c["model"] == "ResHedNet":
        sd = so.init_reshednet(ncls, nf, layers=c["kw"]["layers"], seed=seed)
    else:
        sd = so.init_dilnet(ncls, nf, seed=seed)
    ref = _sd(g, "|init", torch.float32)
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_forward_grads_steps(golden_dir, name, tag, dtype, tol):
    g = _load(golden_dir, name)
    ncls = int(g["meta"][0])
    c = CASES[name]
    sd = _sd(g, "|init", dtype)
    x = torch.from_numpy(g["x"]).to(dtype)
    y = torch.from_numpy(g["y"])
    y = y if ncls > 1 else y.to(dtype)
    opt = so.AdamState(lr=1e-3)
    losses = []
    for s in range(3):
        loss, logits, grads = so.loss_and_grads(c["model"], sd, x, y, ncls, **c["kw"])
        if s == 0:
            np.testing.assert_allclose(logits.numpy(), g["logits|" + tag], rtol=tol, atol=tol)
            gmax = max(float(np.abs(g[k + "|grad|" + tag]).max()) for k in grads)
            for k, gr in grads.items():
                err = float(np.abs(gr.numpy() - g[k + "|grad|" + tag]).max())
                assert err <= 50 * tol * gmax, (k, err, gmax)
            for k in sd:
                if "running" in k:
                    np.testing.assert_allclose(sd[k].numpy(), g[k + "|bn1|" + tag], rtol=tol, atol=tol)
        opt.step(sd, grads)
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g["losses|" + tag], rtol=10 * tol)
    if tag == "f64":       # parameters after optimisation are only meaningful to compare in fp64
        for k in so.param_keys(sd):
            np.testing.assert_allclose(sd[k].numpy(), g[k + "|after|f64"], rtol=1e-7, atol=1e-9)
    ev = so.net_forward(c["model"], OrderedDict(sd), x, False, **c["kw"])
    ref_ev = g["eval_logits|" + tag]
    etol = tol if tag == "f64" else 5e-3   # fp32 params diverge after 3 Adam steps
    np.testing.assert_allclose(ev.detach().numpy(), ref_ev, rtol=etol, atol=etol * np.abs(ref_ev).max())


def test_default_init_moments(golden_dir):
    g = _load(golden_dir, "seg_default_init_moments")
    for model, fn, ncls in (("Unet", so.init_unet, 3), ("dilnet", so.init_dilnet, 1)):
        sd = fn(ncls, seed=1) if model == "Unet" else fn(ncls, seed=1)
        assert sum(sd[k].numel() for k in so.param_keys(sd)) == int(g[f"{model}|nparams"])
        for k, v in sd.items():
            v = v.double()
            ref = g[f"{model}|{k}"]
            got = np.array([v.numel(), v.sum().item(), (v * v).sum().item(),
                            v.flatten()[0].item(), v.flatten()[-1].item()])
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12, err_msg=k)


def test_predict_probs(golden_dir):
    g = _load(golden_dir, "seg_predict")
    for model, ncls, f in (("Unet", 3, 8), ("dilnet", 1, 2)):
        sd = OrderedDict((k.split("|sd|")[1], torch.from_numpy(g[k])) for k in g.files
                         if k.startswith(model + "|sd|"))
        x = torch.from_numpy(g[f"fmt{f}"])
        probs = so.predict_probs(model, sd, x, ncls)
        np.testing.assert_allclose(probs.numpy(), g[f"{model}|probs"], rtol=1e-5, atol=1e-6)
