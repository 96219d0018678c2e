"""Shared bodies of the Segmentor parity tests.  The SAME checks run
  * on the CPU through the SIMT emulator build of the kernel sources (`not gpu` tier), and
  * on a real MI355X through libatomai_amd.so (`gpu` tier),
against golden vectors generated from the real reference (tests/golden, oracle/make_golden.py)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from _knobs import set_knob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "seg_unet_c3_nf4_b2_32": ("Unet", dict()),
    "seg_unet_c1_nf4_b2_16_nearest": ("Unet", dict(upsampling="nearest")),
    "seg_unet_dil_c3_nf4_b2_32": ("Unet", dict(with_dilation=True)),
    "seg_dilnet_c1_nf5_b2_32": ("dilnet", dict()),
    "seg_segresnet_c3_nf4_b2_32": ("SegResNet", dict()),
    "seg_segresnet_c1_nf4_b2_16_nearest": ("SegResNet", dict(upsampling="nearest")),
    "seg_reshednet_c3_nf4_b2_32": ("ResHedNet", dict(layers=[2, 2, 2])),
    "seg_reshednet_c3_nf4_b2_22": ("ResHedNet", dict(layers=[1, 1, 1])),
    "seg_reshednet_c1_nf4_b2_22_nearest": ("ResHedNet", dict(upsampling="nearest", layers=[1, 2, 1])),
    # more classes than the register-resident head kernels hold (8): chunked px / re-reading CE kernels (head.hip)
    "seg_unet_c9_nf4_b2_32": ("Unet", dict()),
    "seg_dilnet_c11_nf5_b2_32": ("dilnet", dict()),
}
REL_TOL = 1e-4          # north_star: "within 1e-4 rel fp32"


def relmax(a, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def check_net_case(name, device):
    from atomai_amd.nets import init_fcnn_model
    from atomai_amd.losses_metrics import select_loss
    from atomai_amd.optim import FusedAdam
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ncls, nf, B, H, seed, dil = [int(v) for v in g["meta"]]
    model, kw = CASES[name]
    torch.manual_seed(seed)                                  # set_train_rng(seed) of the reference
    net, meta = init_fcnn_model(model, ncls, nb_filters=nf, **kw)
    for k, v in net.state_dict().items():                    # RNG-order initialisation == reference
        assert np.array_equal(v.numpy(), g[k + "|init"]), k
    assert meta["model"] == model and meta["nb_classes"] == ncls
    net.to(device)
    x = torch.from_numpy(g["x"]).to(device)
    y = torch.from_numpy(g["y"]).to(device)
    crit = select_loss("ce", ncls)
    opt = FusedAdam(net.parameters(), lr=1e-3)
    opt.prepare()
    losses = []
    for s in range(3):
        net.train()
        opt.zero_grad()
        logits = net(x)
        loss = crit(logits, y)
        loss.backward()
        if s == 0:
            assert relmax(logits.detach().cpu().numpy(), g["logits|f64"]) < REL_TOL
            gmax = max(np.abs(g[k + "|grad|f64"]).max() for k, _ in net.named_parameters())
            for k, p in net.named_parameters():
                ref = g[k + "|grad|f64"]
                err = np.abs(p.grad.cpu().numpy() - ref).max() / gmax
                ref32 = np.abs(g[k + "|grad|f32"] - ref).max() / gmax
                # gradients: judged against fp64, normalised by the global gradient scale and relative to
                # the reference's own fp32 noise (SURVEY.md §7 "Parity budget")
                assert err <= max(4 * ref32, 2e-5), (k, err, ref32)
        opt.step()
        if s == 0:
            for k, v in net.state_dict().items():
                if "running" in k:
                    np.testing.assert_allclose(v.cpu().numpy(), g[k + "|bn1|f64"], rtol=REL_TOL, atol=1e-6)
                if "num_batches_tracked" in k:
                    assert int(v) == 1
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses|f64"], rtol=REL_TOL)
    # the optimizer state keeps torch.optim.Adam's format
    st = opt.state_dict()["state"]
    assert set(st[0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st[0]["step"]) == 3
    net.eval()
    with torch.no_grad():
        ev = net(x).cpu().numpy()
    ref = g["eval_logits|f32"]
    assert relmax(ev, ref.astype(np.float64)) < 2e-2        # parameters after Adam steps: loose


def check_many_classes_predict(device, ncls=11):
    """predict path with more than 8 classes: amx_px_fwd mode 1 (probabilities, NHWC) on the chunked kernel against the
    softmax of the module's own logits; SegPredictor end to end (shape, rows sum to one)."""
    import atomai_amd as aoi
    from atomai_amd.nets import init_fcnn_model
    from atomai_amd.nets.fcnn import predict_proba
    torch.manual_seed(3)
    net, _ = init_fcnn_model("Unet", ncls, nb_filters=4)
    net = net.to(device).eval()
    x = torch.rand(2, 1, 32, 32, device=device)
    with torch.no_grad():
        logits = net(x)
    prob = predict_proba(net, x)
    assert tuple(prob.shape) == (2, 32, 32, ncls)
    ref = torch.softmax(logits.double(), 1).permute(0, 2, 3, 1)
    assert float((prob.double() - ref).abs().max()) < 1e-6
    out = aoi.predictors.SegPredictor(net, use_gpu=(device != "cpu"), verbose=False).predict(
        x[:, 0].cpu().numpy(), compute_coords=False)
    assert out.shape == (2, 32, 32, ncls) and np.allclose(out.sum(-1), 1.0, atol=1e-5)


def check_vs_oracle_small(model, ncls, device, nf=4, B=2, H=16, seed=5, **kw):
    """Configurations without a reference golden (e.g. batch_norm=False): logits, loss and every gradient against
    the pinned oracle evaluated in fp64 on the CPU."""
    from oracle import seg_oracle as so
    from atomai_amd.nets import init_fcnn_model
    from atomai_amd.losses_metrics import select_loss
    torch.manual_seed(seed)
    net, _ = init_fcnn_model(model, ncls, nb_filters=nf, **kw)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.rand(B, 1, H, H).astype(np.float32))
    y = (torch.from_numpy(rs.randint(0, ncls, (B, H, H))) if ncls > 1
         else torch.from_numpy((rs.rand(B, 1, H, H) > 0.5).astype(np.float32)))
    okw = {("upsampling" if k == "upsampling" else k): v for k, v in kw.items()}
    assert so.min_abs_preactivation(model, sd, x, **okw) > 1e-5      # inputs clear of the LeakyReLU kink
    net.to(device).train()
    logits = net(x.to(device))
    loss = select_loss("ce", ncls)(logits, y.to(device))
    loss.backward()
    y64 = y if ncls > 1 else y.double()
    ref_loss, ref_logits, ref_grads = so.loss_and_grads(model, so.cast(sd, torch.float64), x.double(), y64, ncls, **okw)
    assert relmax(logits.detach().cpu().numpy(), ref_logits.numpy()) < REL_TOL
    assert abs(loss.item() - float(ref_loss)) / abs(float(ref_loss)) < 1e-5
    gmax = max(float(g.abs().max()) for g in ref_grads.values())
    for k, p in net.named_parameters():
        err = float((p.grad.cpu().double() - ref_grads[k]).abs().max()) / gmax
        assert err < 2e-5, (k, err)


def check_blocks(device):
    from atomai_amd.nets import ConvBlock, DilatedBlock, UpsampleBlock
    g = np.load(os.path.join(GOLD, "seg_blocks.npz"))
    ctors = {
        "convblock_bn": lambda: ConvBlock(2, 2, 6, 8, batch_norm=True),
        "convblock_nobn_a01": lambda: ConvBlock(2, 2, 1, 8, lrelu_a=0.1),
        "up_bilinear": lambda: UpsampleBlock(2, 8, 4, mode="bilinear"),
        "up_nearest": lambda: UpsampleBlock(2, 8, 4, mode="nearest"),
        "dilated_bn": lambda: DilatedBlock(2, 6, 8, [2, 4, 6], [2, 4, 6], batch_norm=True),
    }
    for name, ctor in ctors.items():
        m = ctor()
        sd = OrderedDict((k.split("|sd|")[1], torch.from_numpy(g[k])) for k in g.files
                         if k.startswith(name + "|sd|"))
        assert list(sd.keys()) == list(m.state_dict().keys()), name       # module tree / key names
        m.load_state_dict(sd)
        m.to(device)
        for mode in ("train", "eval"):
            # (the golden's eval pass ran after its training pass, i.e. with updated running statistics)
            m.train(mode == "train")
            x = torch.from_numpy(g[f"{name}|x"]).to(device).requires_grad_(True)
            y = m(x)
            assert relmax(y.detach().cpu().numpy(), g[f"{name}|y|{mode}|f64"]) < REL_TOL, (name, mode)
            if mode == "eval" and "bn" in name:
                continue        # backward through eval-mode BN is not on the hot path
            m.zero_grad()
            y.backward(torch.from_numpy(g[f"{name}|gy"]).to(device))
            refx = g[f"{name}|gx|{mode}|f64"]
            scale = max(np.abs(refx).max(), 1e-30)
            e = np.abs(x.grad.cpu().numpy() - refx).max() / scale
            e32 = np.abs(g[f"{name}|gx|{mode}|f32"] - refx).max() / scale
            assert e <= max(4 * e32, 2e-5), (name, mode, "gx", e, e32)
            for k, p in m.named_parameters():
                ref = g[f"{name}|gp|{k}|{mode}|f64"]
                sc = max(np.abs(ref).max(), 1e-30)
                e = np.abs(p.grad.cpu().numpy() - ref).max() / sc
                e32 = np.abs(g[f"{name}|gp|{k}|{mode}|f32"] - ref).max() / sc
                assert e <= max(4 * e32, 5e-5), (name, mode, k, e, e32)


def check_dilated_ragged(device, cases=((28, 50, 37, 29, 2), (16, 20, 23, 41, 1), (8, 8, 4, 6, 1), (4, 20, 3, 2, 2)),
                          lattice=None):
    """Dilations 2 / 4 / 6 on image sizes that are multiples of neither the dilation nor the tile (ragged residue
    classes: sub-images of different sizes, empty statistics strips; images SMALLER than the dilation, where some
    residue classes hold no pixel at all), channel counts with a partial last chunk;
    forward, BatchNorm statistics, data and weight gradients against the same layers in stock torch fp64."""
    import copy
    import torch.nn as nn
    from atomai_amd.nets import DilatedBlock
    for cin, cout, H, W, N, *rest in cases:
        # a case may name its LeakyReLU slope: with ~1e6 pre-activations per layer a few land within fp32 rounding of
        # the kink and take the other branch than the fp64 graph does (expected count ~ 0.8e-6 per element); slope 1.0
        # keeps the large-geometry cases a pure index / layout check
        slope = rest[0] if rest else 0.01
        torch.manual_seed(H)
        m = DilatedBlock(2, cin, cout, [2, 4, 6], [2, 4, 6], batch_norm=True, lrelu_a=slope)
        ref_layers = [copy.deepcopy(l).double() for l in m.atrous_module]
        m.to(device)
        x = torch.randn(N, cin, H, W)
        x1 = x.clone().to(device).requires_grad_(True)
        x2 = x.double().clone().requires_grad_(True)
        y = m(x1)
        h, outs = x2, []
        for l in ref_layers:
            h = l(h)
            outs.append(h)                   # the block returns the sum of EVERY sub-layer's output (blocks.py:321-329)
        yr = sum(outs)
        gy = torch.randn(N, cout, H, W)
        y.backward(gy.to(device))
        yr.backward(gy.double())
        assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-4, (cin, cout, H, W)
        assert float((x1.grad.cpu().double() - x2.grad).abs().max() / x2.grad.abs().max()) < 1e-4
        ref_params = [p for l in ref_layers for p in l.parameters()]
        for (k, p), p2 in zip(m.atrous_module.named_parameters(), ref_params):
            assert float((p.grad.cpu().double() - p2.grad).abs().max() / p2.grad.abs().max()) < 1e-4, k
        ref_bn = [l for l in ref_layers if isinstance(l, nn.BatchNorm2d)]
        our_bn = [l for l in m.atrous_module if isinstance(l, nn.BatchNorm2d)]
        for a, b in zip(our_bn, ref_bn):
            np.testing.assert_allclose(a.running_var.cpu().numpy(), b.running_var.numpy(), rtol=1e-5)
            np.testing.assert_allclose(a.running_mean.cpu().numpy(), b.running_mean.numpy(), rtol=1e-4, atol=1e-6)


def check_head_fusion(device, wide=True):
    """Eval mode: the final 1x1 convolution (+ sigmoid / softmax) evaluated in the epilogue of the last 3x3 layer must
    agree with the separate head kernel (different summation order: 1e-6), for probabilities and for raw logits."""
    import atomai_amd as aoi
    from atomai_amd import engine
    from atomai_amd.nets.fcnn import predict_proba
    rs = np.random.RandomState(4)
    made = []
    orig = engine.HeadNode.__init__

    def counting(self, *a, **k):
        made.append(1)
        return orig(self, *a, **k)
    engine.HeadNode.__init__ = counting
    try:
        for name, ncls, kw, hw in (("Unet", 3, dict(nb_filters=4), (24, 40)), ("dilnet", 1, dict(nb_filters=5), (22, 38)),
                                   ("Unet", 1, dict(nb_filters=16 if wide else 6, batch_norm=False), (16, 16)),
                                   ("dilnet", 2, dict(nb_filters=25 if wide else 9), (12, 20))):
            torch.manual_seed(6)
            net, _ = aoi.nets.init_fcnn_model(name, ncls, **kw)
            net = net.to(device)
            net.train()
            with torch.no_grad():                            # non-trivial running statistics
                net(torch.from_numpy(rs.rand(2, 1, *hw).astype(np.float32)).to(device))
            net.eval()
            x = torch.from_numpy(rs.rand(3, 1, *hw).astype(np.float32)).to(device)
            res = {}
            for fuse in (True, False):
                engine.FUSE_HEAD = fuse
                n0 = len(made)
                with torch.no_grad():
                    res[fuse] = (predict_proba(net, x), net(x))
                assert (len(made) - n0 == 2) == fuse, (name, fuse)
            for a, b in zip(res[True], res[False]):
                assert a.shape == b.shape
                assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max())), name
    finally:
        engine.HeadNode.__init__ = orig
        engine.FUSE_HEAD = True


def check_dsum_fusion(device):
    """Eval mode: the sum of a DilatedBlock evaluated in the epilogue of its last layer is BIT-IDENTICAL to the separate
    sum kernel (same terms in the same order), with and without BatchNorm, for 2- and 3-layer blocks."""
    from atomai_amd import engine
    from atomai_amd.nets import DilatedBlock
    rs = np.random.RandomState(8)
    made = []
    orig = engine.DsumConvNode.__init__

    def counting(self, *a, **k):
        made.append(1)
        return orig(self, *a, **k)
    engine.DsumConvNode.__init__ = counting
    try:
        for cin, cout, dils, bn, hw in ((25, 50, [2, 4, 6], True, (22, 38)), (50, 50, [2, 4], True, (16, 24)),
                                        (20, 28, [2, 4, 6], False, (13, 21))):
            torch.manual_seed(9)
            m = DilatedBlock(2, cin, cout, dils, dils, batch_norm=bn).to(device)
            m.train()
            with torch.no_grad():
                m(torch.from_numpy(rs.randn(2, cin, *hw).astype(np.float32)).to(device))
            m.eval()
            x = torch.from_numpy(rs.randn(3, cin, *hw).astype(np.float32)).to(device)
            res = {}
            for fuse in (True, False):
                engine.FUSE_HEAD = fuse
                n0 = len(made)
                with torch.no_grad():
                    res[fuse] = m(x)
                assert (len(made) - n0 == 1) == fuse, (cin, cout, fuse)
            assert torch.equal(res[True], res[False]), (cin, cout, float((res[True] - res[False]).abs().max()))
    finally:
        engine.DsumConvNode.__init__ = orig
        engine.FUSE_HEAD = True


def check_input_norm_fusion(device):
    """The predictor's stack normalisation applied inside the first-layer kernel (Unet / dilnet) or by the separate
    pass (nets with another first layer) gives bit-identical probabilities to normalising first."""
    import atomai_amd as aoi
    from atomai_amd.nets.fcnn import predict_proba, _fuses_input_norm
    rs = np.random.RandomState(2)
    x = torch.from_numpy((rs.rand(3, 1, 24, 40) * 37 - 5).astype(np.float32)).to(device)
    mn, ptp = np.float32(x.min().item()), np.float32((x.max() - x.min()).item())
    from atomai_amd import _lib as L
    xn = torch.empty_like(x)                                 # the predictor's separate normalisation pass
    L.call("amx_sub_div", L.ptr(x), L.ptr(xn), x.numel(), float(mn), float(ptp), L.stream_ptr(x))
    for name, ncls, kw, fused in (("Unet", 3, dict(nb_filters=4), True), ("dilnet", 1, dict(nb_filters=5), True),
                                  ("SegResNet", 2, dict(nb_filters=4), False)):
        torch.manual_seed(5)
        net, _ = aoi.nets.init_fcnn_model(name, ncls, **kw)
        net = net.to(device).eval()
        assert _fuses_input_norm(net, x) == fused, name
        a = predict_proba(net, x, input_norm=(mn, ptp))
        b = predict_proba(net, xn)
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


def check_upsample_exact(device):
    """amx_upsample2x_fwd against a float32 numpy restatement of F.interpolate(scale_factor=2, align_corners=False)
    (blocks.py:130-131) with ATen's association order — exact equality, odd and even low-res heights (a thread serves two
    low-res rows), channel-group counts that do not divide 256, both modes."""
    from atomai_amd import _lib as L
    rs = np.random.RandomState(8)
    f = np.float32

    def taps(o, n, mode):
        i = o >> 1
        if mode == 1:
            return i, i, f(1), f(0)
        if o & 1:
            return i, min(i + 1, n - 1), f(0.75), f(0.25)
        return max(i - 1, 0), i, f(0.25), f(0.75)
    for (N, h, w, Cs) in ((2, 5, 7, 28), (1, 8, 6, 16), (3, 1, 3, 4), (1, 3, 70, 12)):
        v = rs.randn(N, h, w, Cs).astype(np.float32)
        for mode in (0, 1):
            want = np.empty((N, 2 * h, 2 * w, Cs), dtype=np.float32)
            for y in range(2 * h):
                y0, y1, wy0, wy1 = taps(y, h, mode)
                for x in range(2 * w):
                    x0, x1, wx0, wx1 = taps(x, w, mode)
                    if mode == 1:
                        want[:, y, x] = v[:, y0, x0]
                    else:
                        want[:, y, x] = wy0 * (wx0 * v[:, y0, x0] + wx1 * v[:, y0, x1]) + wy1 * (wx0 * v[:, y1, x0] + wx1 * v[:, y1, x1])
            vt = torch.from_numpy(v).to(device)
            u = torch.empty(N, 2 * h, 2 * w, Cs, device=device)
            L.call("amx_upsample2x_fwd", L.ptr(vt), L.ptr(u), N, h, w, Cs, mode, L.stream_ptr(vt))
            assert np.array_equal(u.cpu().numpy(), want), (N, h, w, Cs, mode, np.abs(u.cpu().numpy() - want).max())


def check_pool_fusion(device):
    """Eval mode: the 2x2 max-pool behind a one-layer first block comes out of the first-layer kernel
    (amx_conv1_fwd_pool) — bit-identical to the separate amx_pool2x2_fwd launch, with and without the predictor's input
    normalisation, over several blocks of pixels; shapes the fused kernel does not serve fall back to the separate launch."""
    import atomai_amd as aoi
    from atomai_amd import engine, _lib as L
    from atomai_amd.nets.fcnn import predict_proba
    rs = np.random.RandomState(4)
    calls = []
    orig = L.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    for name, ncls, kw in (("Unet", 3, dict(nb_filters=4)), ("dilnet", 1, dict(nb_filters=5))):
        torch.manual_seed(6)
        net, _ = aoi.nets.init_fcnn_model(name, ncls, **kw)
        net = net.to(device).eval()
        with torch.no_grad():                                  # running statistics away from their (0, 1) initial values
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.add_(torch.from_numpy(rs.randn(m.num_features).astype(np.float32) * 0.1).to(device))
                    m.running_var.mul_(torch.from_numpy((0.5 + rs.rand(m.num_features)).astype(np.float32)).to(device))
                    m.weight.mul_(torch.from_numpy(np.where(rs.rand(m.num_features) > 0.5, 1.0, -1.0).astype(np.float32)).to(device))
        for (B, H, W), served in (((3, 48, 32), True), ((2, 32, 64), True), ((1, 24, 40), False)):
            x = torch.from_numpy((rs.rand(B, 1, H, W) * 9 - 2).astype(np.float32)).to(device)
            for norm in (None, (np.float32(-2.0), np.float32(9.0))):
                outs = []
                for fuse in (True, False):
                    engine.FUSE_POOL = fuse
                    calls.clear()
                    L.call = spy
                    engine.L.call = spy
                    try:
                        outs.append(predict_proba(net, x, input_norm=norm) if norm else predict_proba(net, x))
                    finally:
                        L.call = orig
                        engine.L.call = orig
                        engine.FUSE_POOL = True
                    assert ("amx_conv1_fwd_pool" in calls) == (fuse and served), (name, H, W, fuse, calls[:6])
                assert torch.equal(outs[0], outs[1]), (name, H, W, float((outs[0] - outs[1]).abs().max()))


def check_predict(device_is_gpu):
    import atomai_amd as aoi
    from atomai_amd.utils import img_pad, torch_format_image
    g = np.load(os.path.join(GOLD, "seg_predict.npz"))
    img = g["img"]
    for f in (2, 8):
        assert np.array_equal(img_pad(img.copy(), f), g[f"pad{f}"])
        np.testing.assert_array_equal(torch_format_image(img_pad(img.copy(), f)).numpy(), g[f"fmt{f}"])
    for model, ncls, nf in (("Unet", 3, 4), ("dilnet", 1, 5)):
        torch.manual_seed(1)
        net, _ = aoi.nets.init_fcnn_model(model, ncls, nb_filters=nf)
        sd = OrderedDict((k.split("|sd|")[1], torch.from_numpy(g[k])) for k in g.files
                         if k.startswith(model + "|sd|"))
        net.load_state_dict(sd)
        p = aoi.predictors.SegPredictor(net, use_gpu=device_is_gpu, nb_classes=ncls, verbose=False)
        assert p.downsampling == (8 if model == "Unet" else 2)           # via hooks + mock forward
        probs = p.run(img, compute_coords=False, num_batches=2)
        assert probs.shape == g[f"{model}|probs"].shape
        np.testing.assert_allclose(probs, g[f"{model}|probs"], rtol=REL_TOL, atol=1e-6)
        # float32 stack that needs no padding: normalisation happens on the device after the upload and must be
        # bit-identical to the reference's numpy float32 `(x - min) / ptp` done on the host
        sub = np.ascontiguousarray(img[:, :8, :16])
        probs_dev = p.run(sub, compute_coords=False)
        assert p._norm is not None
        x_host = torch_format_image(sub.copy())
        p._norm = None
        probs_host = p.batch_predict(x_host, probs_dev.shape, 1).numpy()
        assert np.array_equal(probs_dev, probs_host)


def check_wave_specialised_conv(device, cin, cout, monkeypatch, hw=32, batch=2):
    """conv_ws.hip (producer / consumer waves in persistent workgroups) on a two-layer ConvBlock (the second layer reads
    the first through its BatchNorm affine) and, for cin == 32, a two-source layer (U-Net's skip | upsampled concat):
    the launches must be taken by the specialised kernel, agree with the general kernel to fp32 summation-order noise
    and with fp64 autograd to 1e-4 — forward, statistics (through BatchNorm), input and parameter gradients."""
    import copy
    import torch.nn as nn
    from atomai_amd import _lib as L
    from atomai_amd.nets import ConvBlock
    out = {}
    for ws in ("1", "0"):
        set_knob(monkeypatch, "AMX_CONV_WS", ws)            # 1 + every data-gradient class (the default mask leaves one out)
        set_knob(monkeypatch, "AMX_CONV_WS_DGRAD", "7")
        torch.manual_seed(3)
        m = ConvBlock(2, 2, cin, cout, batch_norm=True).to(device)
        ref = nn.Sequential(*[copy.deepcopy(l) for l in m.block]).double()
        x = torch.randn(batch, cin, hw, hw, device=device)
        x1, x2 = x.clone().requires_grad_(True), x.double().clone().requires_grad_(True)
        n0 = L.load().amx_conv2d_ws_launches()
        y, yr = m(x1), ref(x2)
        gy = torch.randn_like(y)
        y.backward(gy)
        yr.backward(gy.double())
        n1 = L.load().amx_conv2d_ws_launches()
        # forward of both layers + the data gradient of the second (the first layer's input gradient too when asked for)
        assert (n1 - n0 >= 3) == (ws == "1"), (ws, n1 - n0)
        errs = [float((y.detach().double() - yr.detach()).abs().max()),
                float((x1.grad.double() - x2.grad).abs().max() / x2.grad.abs().max())]
        errs += [float((p.grad.double() - p2.grad).abs().max() / p2.grad.abs().max())
                 for p, p2 in zip(m.block.parameters(), ref.parameters())]
        out[ws] = (y.detach().cpu(), x1.grad.cpu(), [p.grad.cpu() for p in m.parameters()], errs)
    # the output against fp64 autograd: 1e-4; every gradient: the specialised kernel no further from fp64 than the general
    # one (the input gradient through two training-mode BatchNorms is cancellation-dominated at large sizes, so its fp32
    # noise floor — identical for both kernels — is what the second criterion is relative to)
    assert out["1"][3][0] < 1e-4 and out["0"][3][0] < 1e-4, (out["1"][3], out["0"][3])
    assert all(a < 1.5 * b + 2e-6 for a, b in zip(out["1"][3], out["0"][3])), (out["1"][3], out["0"][3])
    assert float((out["1"][0] - out["0"][0]).abs().max()) < 2e-5
    assert float((out["1"][1] - out["0"][1]).abs().max()) < 2e-5 * max(1.0, float(out["0"][1].abs().max()))
    for a, b in zip(out["1"][2], out["0"][2]):
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))


def check_wave_specialised_concat(device, monkeypatch, hw=32, batch=2):
    """U-Net's last block on conv_ws.hip: a layer reading torch.cat([skip, upsampled], 1) from two 16-channel sources,
    each through its own BatchNorm affine, and its data gradient written to two outputs — against the general kernel."""
    import atomai_amd as aoi
    from atomai_amd import _lib as L
    out = {}
    for ws in ("1", "0"):
        set_knob(monkeypatch, "AMX_CONV_WS", ws)
        set_knob(monkeypatch, "AMX_CONV_WS_DGRAD", "7")
        torch.manual_seed(5)
        net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=16)
        net = net.to(device).train()
        x = torch.randn(batch, 1, hw, hw, device=device)
        n0 = L.load().amx_conv2d_ws_launches()
        y = net(x)
        y.backward(torch.ones_like(y) / y.numel())
        assert (L.load().amx_conv2d_ws_launches() - n0 > 0) == (ws == "1")
        out[ws] = [y.detach().cpu()] + [p.grad.cpu() for p in net.parameters()]
    for a, b in zip(out["1"], out["0"]):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), float((a - b).abs().max())


def check_loss_upstream_gradient(device):
    """loss.backward() multiplies the logits gradient by the upstream gradient of the scalar loss: 1.0 takes the
    no-pass shortcut (amx_scale_unless_one), anything else scales in place — both against torch's own losses in fp64
    (reference: trainers/trainer.py:203-206 calls loss.backward() on criterion(prob, y))."""
    import torch.nn.functional as F
    from atomai_amd.losses_metrics import select_loss
    rs = np.random.RandomState(3)
    for ncls in (3, 1):
        x = torch.from_numpy(rs.randn(2, ncls, 12, 20).astype(np.float32))
        if ncls == 1:
            y = torch.from_numpy((rs.rand(2, 1, 12, 20) > 0.5).astype(np.float32))
        else:
            y = torch.from_numpy(rs.randint(0, ncls, (2, 12, 20)))
        for factor in (1.0, 0.37, -2.0):
            xd = x.clone().to(device).requires_grad_(True)
            (select_loss("ce", ncls)(xd, y.to(device)) * factor).backward()
            xr = x.double().requires_grad_(True)
            ref = F.cross_entropy(xr, y) if ncls > 1 else F.binary_cross_entropy_with_logits(xr, y.double())
            (ref * factor).backward()
            assert relmax(xd.grad.cpu().numpy(), xr.grad.numpy()) < 2e-6, (ncls, factor)


def check_wgrad_ws_bit_identical(device, cin, cout, H, N, monkeypatch, W=None, two_sources=None, force_th=None):
    """wgrad_ws.hip against wgrad_kernel.h through the C ABI (amx_conv2d_wgrad_fused): partial rows and bias partials
    bit-identical; the summed rows against torch's conv2d_weight in fp64."""
    from atomai_amd import _lib as L
    W = W or H + 7
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    two = (cin >= 32) if two_sources is None else two_sources
    c0 = cin // 2 if two else cin
    c1 = cin - c0
    x0 = torch.randn(N, H, W, c0, generator=g).to(device)
    x1 = torch.randn(N, H, W, c1, generator=g).to(device) if two else None
    sc0, sh0 = (torch.rand(c0, generator=g) + 0.5).to(device), torch.randn(c0, generator=g).to(device)
    sc1, sh1 = ((torch.rand(c1, generator=g) + 0.5).to(device), torch.randn(c1, generator=g).to(device)) if two else (None, None)
    cos = -(-cout // 4) * 4
    dy = torch.randn(N, H, W, cos, generator=g).to(device)
    aux = torch.randn(N, H, W, cos, generator=g).to(device)
    k1, k2, k3 = ((torch.randn(cos, generator=g) * s).to(device) for s in (1.0, 0.1, 0.05))
    slope = 0.01
    lib = L.load()
    ci_pad, co_pad = -(-cin // 16) * 16, -(-cout // 16) * 16
    res = {}
    if force_th:
        set_knob(monkeypatch, "AMX_WGRAD_TH", str(force_th))
    else:
        set_knob(monkeypatch, "AMX_WGRAD_TH", None)
    set_knob(monkeypatch, "AMX_WGRAD_WS_MASK", "7")           # every class (the product default leaves the 64-channel one out)
    for ws in ("0", "1"):
        set_knob(monkeypatch, "AMX_WGRAD_WS", ws)
        rows = lib.amx_conv2d_wgrad_rows(N, H, W, cin, cout, 9, 1)       # (the plan may pick taller tiles for wgrad_ws.hip)
        ks = lib.amx_conv2d_wgrad_ksplit(N, H, W, cin, cout, 9, 1)
        part = torch.full((rows, 9, ci_pad, co_pad), float("nan"), device=device)
        bpart = torch.full((ks, co_pad), float("nan"), device=device)
        n0 = lib.amx_conv2d_wgrad_ws_launches()
        L.call("amx_conv2d_wgrad_fused", L.ptr(x0), L.ptr(sc0), L.ptr(sh0), c0, L.ptr(x1), L.ptr(sc1), L.ptr(sh1), c1,
               L.ptr(dy), L.ptr(aux), L.ptr(k1), L.ptr(k2), L.ptr(k3), slope, cos, L.ptr(part), L.ptr(bpart),
               N, H, W, cout, 9, 1, L.stream_ptr(x0))
        if device != "cpu":
            torch.cuda.synchronize()
        assert (lib.amx_conv2d_wgrad_ws_launches() - n0 == 1) == (ws == "1")
        res[ws] = (part.cpu(), bpart.cpu())
    assert not torch.isnan(res["1"][0]).any() and not torch.isnan(res["1"][1][:, :cout]).any()
    # (16 -> 32..48 channels: the plan picks 8-row tiles for wgrad_ws.hip and 4-row tiles otherwise unless AMX_WGRAD_TH says)
    if force_th is not None or not (cin <= 16 and 32 <= cout <= 48):
        assert torch.equal(res["0"][0], res["1"][0])
        # (bias partials: the 8 producer waves of wgrad_ws.hip split a tile's pixels differently from the 4 waves of
        #  wgrad_kernel.h — another fixed summation order; they are compared with fp64 below)
    # against fp64 autograd of the convolution
    xin = x0.double().cpu() * sc0.double().cpu() + sh0.double().cpu()
    if two:
        xin = torch.cat([xin, x1.double().cpu() * sc1.double().cpu() + sh1.double().cpu()], -1)
    a64, d64 = aux.double().cpu(), dy.double().cpu()
    dpre = torch.where(a64 > 0, 1.0, slope) * (k1.double().cpu() * d64 + k2.double().cpu() * a64 + k3.double().cpu())
    dw = torch.nn.grad.conv2d_weight(xin.permute(0, 3, 1, 2), (cout, cin, 3, 3), dpre[..., :cout].permute(0, 3, 1, 2),
                                     padding=1)                                            # [co][ci][3][3]
    db = dpre[..., :cout].sum((0, 1, 2))
    for ws in ("0", "1"):
        got = res[ws][0].double().sum(0)[:, :cin, :cout].permute(2, 1, 0).reshape(cout, cin, 3, 3)
        assert float((got - dw).abs().max() / dw.abs().max()) < 2e-5
        assert float((res[ws][1].double().sum(0)[:cout] - db).abs().max() / db.abs().max()) < 2e-5


def check_bwd_fused_in_loaders(device, cin, cout, monkeypatch, hw=32, batch=2, unet=False, res=False, repeats=1):
    """Round 4: BatchNorm / LeakyReLU backward formed inside the loaders of the wave-specialised data-gradient and
    weight-gradient kernels (amx_conv2d_dgrad_fused / amx_conv2d_wgrad_fused) against the two-pass form (amx_bn_bwd_apply
    materialises dpre): the loaders use amx_bn_bwd_apply's arithmetic, so input and weight gradients are BIT-IDENTICAL;
    the conv bias gradient is summed in another (fixed) order.  Also checks that the fused launches happen and that
    amx_bn_bwd_apply is skipped for exactly those layers."""
    import atomai_amd as aoi
    from atomai_amd import _lib as L
    from atomai_amd.nets import ConvBlock, ResModule
    set_knob(monkeypatch, "AMX_CONV_WS", "1")
    set_knob(monkeypatch, "AMX_CONV_WS_DGRAD", "7")
    import atomai_amd.engine as eng
    monkeypatch.setattr(eng, "DGRAD_SPLIT", False)       # (one launch per fused layer here; the two-launch form of a two-source
    #                                                       layer has its own test: check_split_two_source_dgrad)
    out, calls = {}, {}
    real_call = L.call
    # "1" / "0": loaders vs two-pass form with the BatchNorm-backward sums from amx_bn_bwd_reduce in both (bit-identity);
    # "s": the loaders AND the sums of the source layer from the data-gradient kernel's epilogue (round 6, the default)
    for mode in ("1", "0", "s"):
        set_knob(monkeypatch, "AMX_BWD_FUSE", "0" if mode == "0" else "1")
        set_knob(monkeypatch, "AMX_BWD_SUMS", "1" if mode == "s" else "0")
        cnt = {}

        def counting(name, *a, _c=cnt):
            _c[name] = _c.get(name, 0) + 1
            return real_call(name, *a)
        monkeypatch.setattr(L, "call", counting)
        if unet:
            torch.manual_seed(5)
            net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=16)
            net = net.to(device).train()
            x = torch.randn(batch, 1, hw, hw, device=device)
            y = net(x)
            y.backward(torch.ones_like(y) / y.numel())
            grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
        elif res:
            # ResBlock: c2's fused weight gradient reads the block's activation gradient on the SIDE stream
            # while c1's data gradient accumulates into the residual branch's gradient on the main stream — the two must
            # not be one tensor (engine.ResOutNode.backward).  Repeated: a race would not show on every launch.
            torch.manual_seed(3)
            m = ResModule(2, 2, cin, cout, batch_norm=True).to(device)
            x0 = torch.randn(batch, cin, hw, hw, device=device)
            torch.manual_seed(4)
            gy = torch.randn(batch, cout, hw, hw, device=device)
            grads = None
            for _ in range(repeats):
                m.zero_grad()
                x = x0.clone().requires_grad_(True)
                m(x).backward(gy)
                cur = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
                cur["x"] = x.grad.detach().cpu()
                if grads is not None:
                    for k in cur:
                        assert torch.equal(cur[k], grads[k]), ("not reproducible between identical launches", mode, k)
                grads = cur
        else:
            torch.manual_seed(3)
            m = ConvBlock(2, 2, cin, cout, batch_norm=True).to(device)
            x = torch.randn(batch, cin, hw, hw, device=device).requires_grad_(True)
            y = m(x)
            torch.manual_seed(4)
            y.backward(torch.randn_like(y))
            grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
            grads["x"] = x.grad.detach().cpu()
        monkeypatch.setattr(L, "call", real_call)
        out[mode], calls[mode] = grads, cnt
    nf = calls["1"].get("amx_conv2d_dgrad_fused", 0)
    assert nf >= 1 and calls["0"].get("amx_conv2d_dgrad_fused", 0) == 0, calls
    # sums in the epilogue: every such launch replaces one amx_bn_bwd_reduce, results agree to rounding (another summation
    # order of ~1e3..1e6 products per channel), and the same launches stay loader-fused
    nb = calls["s"].get("amx_conv2d_dgrad_fused_bsum", 0)
    assert calls["1"].get("amx_conv2d_dgrad_fused_bsum", 0) == 0
    assert nb + calls["s"].get("amx_conv2d_dgrad_fused", 0) == nf, calls
    assert calls["1"]["amx_bn_bwd_reduce"] - calls["s"].get("amx_bn_bwd_reduce", 0) == nb, calls
    if not res:
        assert nb >= 1, calls
    for k in out["1"]:
        a, b = out["s"][k], out["1"][k]
        tol = 2e-5 * max(1e-3, float(b.abs().max())) + 3e-6 * float(batch * hw * hw) ** 0.5 * (0.0 if a.ndim == 4 else 1.0)
        assert float((a - b).abs().max()) <= tol, ("sums in the data-gradient epilogue", k, float((a - b).abs().max()), tol)
    # (+ the net's first layer, whose BatchNorm backward is formed by amx_conv1_wgrad_fused: it has no data gradient)
    per_run = repeats if res else 1
    # (... or, in front of a pooling layer, by amx_pool2x2_bwd_wgrad1 + amx_conv1_wgrad_combine, which sum the three terms of
    #  that layer's weight gradient separately: equal to rounding, not bit for bit)
    pooled_first = calls["1"].get("amx_pool2x2_bwd_wgrad1", 0)
    assert (calls["0"]["amx_bn_bwd_apply"] - calls["1"].get("amx_bn_bwd_apply", 0)
            == nf + calls["1"].get("amx_conv1_wgrad_fused", 0) + pooled_first), calls
    nf //= per_run
    first_w = next(iter(out["0"])) if pooled_first else None
    for k in out["0"]:
        a, b = out["1"][k], out["0"][k]
        if k == first_w:
            assert float((a - b).abs().max()) <= 2e-5 * max(1e-3, float(b.abs().max())), (k, float((a - b).abs().max()))
        elif k == "x" or (k.endswith("weight") and a.ndim == 4):
            assert torch.equal(a, b), k
        else:
            # (a conv bias in front of a BatchNorm has a mathematically ZERO gradient: what both forms produce is the
            #  rounding residue of a cancelling sum over all pixels, summed in two different fixed orders)
            tol = 2e-5 * max(1.0, float(b.abs().max())) + 3e-6 * float(batch * hw * hw) ** 0.5
            assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)
    return nf


def check_remainder_columns(device, cases=((50, 50, 1, 40, 2), (50, 50, 2, 44, 2), (25, 50, 4, 36, 1), (50, 25, 1, 32, 2),
                                           (28, 50, 6, 37, 1))):
    """conv_kernel.h REM classes (28 = 16 + 3 x 4 and 52 = 3 x 16 + 4 columns, the 4-wide blocks on v_mfma_f32_4x4x1)
    against the padded power-of-two plan (AMX_CONV_REM=0) on one training-mode ConvBlock / DilatedBlock layer: forward,
    BatchNorm statistics (through the running buffers), input and parameter gradients.  The 16-wide column tiles run the
    same MFMA sequence in both plans; the remainder couts sum their channels in another order (fp32 rounding level)."""
    import copy
    from atomai_amd import _lib as L
    from atomai_amd.nets import ConvBlock, DilatedBlock
    for cin, cout, dil, hw, batch in cases:
        res = {}
        for rem in ("1", "0"):
            L.set_knob("AMX_CONV_REM", rem)
            try:
                torch.manual_seed(11)
                m = (ConvBlock(2, 1, cin, cout, batch_norm=True) if dil == 1
                     else DilatedBlock(2, cin, cout, [dil], [dil], batch_norm=True)).to(device)
                x = torch.randn(batch, cin, hw, hw + 5, device=device).requires_grad_(True)
                n0 = L.load().amx_conv2d_rem_launches()
                y = m(x)
                torch.manual_seed(12)
                y.backward(torch.randn_like(y))
                used = L.load().amx_conv2d_rem_launches() - n0
            finally:
                L.set_knob("AMX_CONV_REM", None)
            assert (used >= 1) == (rem == "1"), (cin, cout, dil, rem, used)
            res[rem] = [y.detach().cpu(), x.grad.detach().cpu()] + [p.grad.detach().cpu() for p in m.parameters()] + \
                       [b.detach().cpu().float() for b in m.buffers()]
        for i, (a, b) in enumerate(zip(res["1"], res["0"])):
            tol = 2e-5 * max(1.0, float(b.abs().max()))
            assert float((a - b).abs().max()) <= tol, (cin, cout, dil, i, float((a - b).abs().max()), tol)
        if dil == 1:                                   # ConvBlock output = the layer itself: the first 16-wide tiles agree exactly
            nmain = (-(-cout // 4) * 4 // 16) * 16
            assert torch.equal(res["1"][0][:, :nmain], res["0"][0][:, :nmain]) or cout < 16


def check_upconv_fused_kernel(device, cases=((50, 25, 2, 20, 33, 0), (128, 64, 2, 13, 16, 0), (64, 32, 1, 16, 30, 1),
                                             (32, 16, 3, 7, 5, 0), (20, 8, 2, 15, 29, 0), (3, 5, 1, 6, 14, 1))):
    """amx_upconv1x1_fwd (UpsampleBlock forward in one pass, upconv.hip) against the two-kernel path it replaces —
    amx_conv2d_fwd(taps = 1) with the producer's BatchNorm affine on load, then amx_upsample2x_fwd — BIT for bit, on ragged
    sizes (tiles of 6 x 14 low-res pixels: partial tiles on both axes), both interpolation modes, channel counts that are
    not multiples of 4 / 16; and against torch's own F.interpolate -> Conv2d (the reference's order of the two operations)."""
    import torch.nn.functional as F
    from atomai_amd import _lib as L
    from atomai_amd import engine as E
    for cin, cout, N, h, w, mode in cases:
        torch.manual_seed(cin + cout)
        conv = torch.nn.Conv2d(cin, cout, 1).to(device)
        x = torch.randn(N, cin, h, w, device=device)
        sc = (torch.rand(cin, device=device) + 0.5)
        sh = torch.randn(cin, device=device) * 0.3
        tape = E.Tape(False, False)
        src = tape.input(x).out
        cs_in, cs_out = src.Cs, E.r4(cout)
        scp, shp = E.padded_vec(sc, cs_in), E.padded_vec(sh, cs_in)
        src.scale, src.shift = scp, shp
        v = tape.conv([src], conv, None, 1.0)                  # low-res 1x1 convolution (affine on load)
        ref = tape.upsample(v, "bilinear" if mode == 0 else "nearest").t
        assert L.load().amx_upconv1x1_supported(cin, cs_in, cout, cs_out) == 1
        assert L.load().amx_upconv1x1_supported(6, 8, 8, 8) == 0        # 2-group tail: another summation order -> two kernels
        y = torch.empty((N, 2 * h, 2 * w, cs_out), dtype=torch.float32, device=x.device)
        L.call("amx_upconv1x1_fwd", L.ptr(src.t), L.ptr(scp), L.ptr(shp), L.ptr(conv.weight.detach().contiguous()),
               L.ptr(conv.bias.detach()), L.ptr(y), N, h, w, cin, cs_in, cout, cs_out, mode, L.stream_ptr(y))
        assert torch.equal(y[..., :cout], ref[..., :cout]), (cin, cout, N, h, w, mode,
                                                               float((y[..., :cout] - ref[..., :cout]).abs().max()))
        assert float(y[..., cout:].abs().max()) == 0.0 if cs_out > cout else True
        xa = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        up = F.interpolate(xa.double(), scale_factor=2, mode="bilinear" if mode == 0 else "nearest",
                           **({"align_corners": False} if mode == 0 else {}))
        want = F.conv2d(up, conv.weight.double(), conv.bias.double()).permute(0, 2, 3, 1)
        assert float((y[..., :cout].double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def check_upconv_node(device, hw=32, batch=2):
    """UpsampleBlock as ONE forward launch (engine.UpConvNode) inside whole nets: U-Net (three UpsampleBlocks, BatchNorm
    affine of the producer on load) and dilnet (50 -> 25 channels, no affine) — logits and EVERY gradient bit-identical to
    the UpsampleNode + ConvNode pair (the size threshold is lowered so that the small test shapes take the fused path)."""
    import atomai_amd as aoi
    from atomai_amd import _lib as L
    from atomai_amd import engine as E
    real_call = L.call
    for model, ncls, nf in (("Unet", 3, 16), ("dilnet", 1, 25)):
        out = {}
        for thr in (1 << 40, 0):
            cnt = {}

            def counting(name, *a, _c=cnt):
                _c[name] = _c.get(name, 0) + 1
                return real_call(name, *a)
            prev = E.UPCONV_MIN_PIXELS
            E.UPCONV_MIN_PIXELS = thr
            L.call = counting
            try:
                torch.manual_seed(5)
                net, _ = aoi.nets.init_fcnn_model(model, ncls, nb_filters=nf)
                net = net.to(device).train()
                x = torch.randn(batch, 1, hw, hw, device=device)
                y = net(x)
                y.backward(torch.ones_like(y) / y.numel())
                out[thr] = ([y.detach().cpu()] + [p.grad.detach().cpu() for p in net.parameters()], cnt)
            finally:
                L.call = real_call
                E.UPCONV_MIN_PIXELS = prev
        n_up = 3 if model == "Unet" else 1
        assert out[0][1].get("amx_upconv1x1_fwd", 0) == n_up and out[1 << 40][1].get("amx_upconv1x1_fwd", 0) == 0, model
        assert out[1 << 40][1]["amx_upsample2x_fwd"] - out[0][1].get("amx_upsample2x_fwd", 0) == n_up
        assert out[0][1]["amx_upsample2x_bwd"] == out[1 << 40][1]["amx_upsample2x_bwd"] == n_up
        for i, (a, b) in enumerate(zip(out[0][0], out[1 << 40][0])):
            assert torch.equal(a, b), (model, i, float((a - b).abs().max()))



def check_hooked_forward_equals_fused(device, models=(("Unet", 3), ("dilnet", 1), ("SegResNet", 3), ("ResHedNet", 1))):
    """A forward hook on any block switches the nets to the block-by-block path (utils/nn.py get_downsample_factor relies on
    it); pooling and the final 1x1 convolution of that path run on the HIP nodes as well.  Same logits as
    the fused single-tape path, and gradients reach every parameter."""
    import warnings
    from atomai_amd.nets import init_fcnn_model
    for model, ncls in models:
        torch.manual_seed(3)
        net, _ = init_fcnn_model(model, ncls, nb_filters=4)
        net.to(device).train()
        x = torch.from_numpy(np.random.RandomState(3).rand(2, 1, 16, 16).astype(np.float32)).to(device)
        ref = net(x)
        seen = []
        first = next(iter(net.children()))
        h = first.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for m in net.modules():                       # same batch statistics again: running stats do not matter
                    if isinstance(m, torch.nn.BatchNorm2d):
                        m.momentum = 0.0
                out = net(x)
        finally:
            h.remove()
        assert seen, model
        assert float((out - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), model
        out.sum().backward()
        assert all(p.grad is not None for p in net.parameters()), model


def check_lattice_xpack_bit_identical(device, monkeypatch, cases=((52, 50, 44, 70, 2, 6), (28, 50, 37, 29, 1, 4), (52, 50, 64, 512, 1, 6))):
    """x-packed lattice tiles (conv_kernel.h, ConvFwdArgs::xpack: the sub-images of a residue row side by side on one tile
    axis) against one sub-image per tile axis: the eval-mode forward of a dilated layer and its data gradient — the launches
    that write no batch statistics — must be BIT-IDENTICAL, and the packed plan must actually be the one taken."""
    import torch.nn as nn
    from atomai_amd import _lib as L
    from atomai_amd.engine import Tape
    for cin, cout, H, W, N, dil in cases:
        assert L.load().amx_conv2d_lattice_xpack(W, dil, 0) > 0, (W, dil)
        assert L.load().amx_conv2d_lattice_xpack(W, dil, 1) == 0
        torch.manual_seed(W)
        conv = nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil).to(device)
        bn = nn.BatchNorm2d(cout).to(device)
        with torch.no_grad():
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 1.5)
        x = torch.randn(N, cin, H, W, device=device)
        gy = torch.randn(N, cout, H, W, device=device)
        ys, gs = [], []
        for knob in ("0", "1"):
            L.set_knob("AMX_CONV_XPACK", knob)
            tape = Tape(False, False)                            # eval forward: no statistics
            ys.append(tape.output(tape.conv([tape.input(x).out], conv, bn, 0.01)).value.clone())
        for knob in ("0", "1"):                                  # (after the eval passes: training updates the running statistics)
            L.set_knob("AMX_CONV_XPACK", knob)
            xg = x.clone().requires_grad_(True)
            tape = Tape(True, True)                              # training: the data gradient has no statistics either
            n = tape.input(xg)
            o = tape.output(tape.conv([n.out], conv, None, 0.01))
            o.grad_out = gy
            tape.backward()
            gs.append(n.grad_nchw.clone())
        L.set_knob("AMX_CONV_XPACK", None)
        assert torch.equal(ys[0], ys[1]), (cin, cout, H, W, dil, "forward")
        assert torch.equal(gs[0], gs[1]), (cin, cout, H, W, dil, "dgrad")


def check_split_two_source_dgrad(device, monkeypatch, hw=32, batch=2):
    """Round 6: the data gradient of a layer that read a concatenation of two 32-channel sources (U-Net c5.0) runs as two
    launches of the wave-specialised kernel, one per source, each with the BatchNorm / LeakyReLU backward formed by its
    loader and the weight image of its half (amx_pack_weights_range) — against the single general-kernel launch behind
    amx_bn_bwd_apply: both source gradients and the weight gradient BIT-IDENTICAL (same MFMA order per output, same
    dpre arithmetic), the layer's amx_bn_bwd_apply pass gone."""
    import atomai_amd.engine as eng
    from atomai_amd import _lib as L
    import torch.nn as nn
    from atomai_amd.engine import Tape
    set_knob(monkeypatch, "AMX_CONV_WS", "1")
    set_knob(monkeypatch, "AMX_CONV_WS_DGRAD", "7")
    set_knob(monkeypatch, "AMX_BWD_FUSE", "1")
    torch.manual_seed(11)
    conv = nn.Conv2d(64, 32, 3, padding=1).to(device)
    bn = nn.BatchNorm2d(32).to(device)
    xs = [torch.randn(batch, 32, hw, hw, device=device) for _ in range(2)]
    gy = torch.randn(batch, 32, hw, hw, device=device)
    res, calls = {}, {}
    real_call = L.call
    for split in (True, False):
        monkeypatch.setattr(eng, "DGRAD_SPLIT", split)
        cnt = {}

        def counting(name, *a, _c=cnt):
            _c[name] = _c.get(name, 0) + 1
            return real_call(name, *a)
        monkeypatch.setattr(L, "call", counting)
        xin = [x.clone().requires_grad_(True) for x in xs]
        tape = Tape(True, True)
        ins = [tape.input(x) for x in xin]
        o = tape.output(tape.conv([n.out for n in ins], conv, bn, 0.01))
        o.grad_out = gy
        tape.backward()
        monkeypatch.setattr(L, "call", real_call)
        res[split] = [n.grad_nchw.clone() for n in ins] + [tape.param_grads[id(conv.weight)][1].clone()]
        calls[split] = cnt
    assert calls[True].get("amx_conv2d_dgrad_fused", 0) == 2 and calls[True].get("amx_bn_bwd_apply", 0) == 0, calls
    assert calls[False].get("amx_conv2d_dgrad_fused", 0) == 0 and calls[False].get("amx_bn_bwd_apply", 0) == 1, calls
    for a, b, nm in zip(res[True], res[False], ("dx0", "dx1", "dW")):
        assert torch.equal(a, b), nm


def check_fused_head_and_loss(device, models=(("Unet", 3, 4), ("Unet", 2, 8), ("SegResNet", 3, 4), ("ResHedNet", 3, 4), ("dilnet", 4, 8),
                                              ("Unet", 1, 4), ("dilnet", 1, 8))):
    """net.forward_loss (px -> CrossEntropyLoss -> their backward as one pass, engine.PxLossNode) against net(x) +
    criterion + backward: the same loss and the same gradient of every parameter and of the input (fp32 rounding apart), with
    an upstream gradient of 1 and of 0.37; nets whose last activation the fused kernel does not take fall back to logits."""
    from atomai_amd.losses_metrics.losses import select_loss
    from atomai_amd.nets import init_fcnn_model
    rs = np.random.RandomState(7)
    kinds = []
    for model, ncls, nf in models:
        crit = select_loss("ce", ncls if ncls != 2 else 3)      # (CrossEntropyLoss for >= 2 classes, BCEWithLogitsLoss for 1)
        for gscale in (1.0, 0.37):
            torch.manual_seed(5)
            net, _ = init_fcnn_model(model, ncls, nb_filters=nf)
            net.to(device).train()
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.momentum = 0.0                           # (two forwards over the same batch)
            x = torch.from_numpy(rs.rand(3, 1, 24, 32).astype(np.float32)).to(device).requires_grad_(True)
            if ncls == 1:
                y = torch.from_numpy((rs.rand(3, 1, 24, 32) > 0.6).astype(np.float32)).to(device)
            else:
                y = torch.from_numpy(rs.randint(0, ncls, (3, 24, 32))).to(device)
            loss0 = crit(net(x), y)
            (loss0 * gscale).backward()
            ref = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
            net.zero_grad()
            x.grad = None
            kind, out = net.forward_loss(x, y)
            kinds.append(kind)
            loss1 = out if kind == "loss" else crit(out, y)
            (loss1 * gscale).backward()
            got = [p.grad for p in net.parameters()] + [x.grad]
            assert abs(float(loss0) - float(loss1)) < 2e-6 * max(1.0, abs(float(loss0))), (model, float(loss0), float(loss1))
            gmax = max(float(g.abs().max()) for g in ref)
            for (name, _), a, b in zip(list(net.named_parameters()) + [("input", None)], got, ref):
                assert float((a - b).abs().max()) < 2e-5 * gmax, (model, ncls, name, float((a - b).abs().max()), gmax)
    assert kinds.count("loss") >= 8, kinds                    # U-Net (16-channel-class heads), SegResNet, ... take the fused node
    net.eval()
    assert net.forward_loss(x, y)[0] == "logits"


def check_pool_backward_with_first_layer_wgrad(device, models=(("Unet", 3, 4), ("Unet", 1, 16), ("dilnet", 1, 8), ("SegResNet", 3, 4))):
    """amx_pool2x2_bwd_wgrad1 (pooling backward + the first layer's weight-gradient sums in one pass, dy never written)
    against the two kernels it replaces: every parameter gradient of the net agrees to fp32 rounding, the first layer's
    included, and the fused kernel is the one that ran."""
    import atomai_amd.engine as eng
    from atomai_amd import _lib as L
    from atomai_amd.losses_metrics.losses import select_loss
    from atomai_amd.nets import init_fcnn_model
    rs = np.random.RandomState(11)
    for model, ncls, nf in models:
        crit = select_loss("ce", ncls if ncls != 2 else 3)
        x = torch.from_numpy(rs.rand(3, 1, 24, 32).astype(np.float32)).to(device)
        y = (torch.from_numpy((rs.rand(3, 1, 24, 32) > 0.6).astype(np.float32)) if ncls == 1
             else torch.from_numpy(rs.randint(0, ncls, (3, 24, 32)))).to(device)
        grads, calls = [], []
        orig = L.call
        for on in (False, True):
            eng.FUSE_POOL_WGRAD1 = on
            seen = []

            def spy(name, *a, _seen=seen):
                _seen.append(name)
                return orig(name, *a)
            L.call = eng.L.call = spy
            try:
                torch.manual_seed(5)
                net, _ = init_fcnn_model(model, ncls, nb_filters=nf)
                net.to(device).train()
                crit(net(x), y).backward()
            finally:
                L.call = eng.L.call = orig
                eng.FUSE_POOL_WGRAD1 = True
            grads.append([p.grad.clone() for p in net.parameters()])
            calls.append(seen)
        assert "amx_pool2x2_bwd_wgrad1" in calls[1] and "amx_conv1_wgrad_fused" not in calls[1], model
        assert "amx_pool2x2_bwd_wgrad1" not in calls[0] and "amx_conv1_wgrad_fused" in calls[0], model
        gmax = max(float(g.abs().max()) for g in grads[0])
        for (name, _), a, b in zip(net.named_parameters(), grads[1], grads[0]):
            assert float((a - b).abs().max()) < 2e-5 * gmax, (model, name, float((a - b).abs().max()), gmax)
