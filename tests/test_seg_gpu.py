"""`gpu` tier: the parity tests proper, through the C ABI of libatomai_amd.so on a real MI355X."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import _seg_checks as C

pytestmark = pytest.mark.gpu
GOLD = C.GOLD


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "gpu tier needs an MI355X"
    from atomai_amd import _lib
    _lib.load()                                   # raises if the HIP extension is missing
    assert not _lib.is_test_backend()
    maps = open("/proc/self/maps").read()
    assert "libatomai_amd.so" in maps, "native library not mapped"
    hips = {l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l}
    assert len(hips) == 1, f"more than one HIP runtime mapped: {hips}"


@pytest.mark.parametrize("name", list(C.CASES))
def test_net_fwd_bwd_adam(name):
    C.check_net_case(name, "cuda")


def test_blocks():
    C.check_blocks("cuda")


def test_dilated_layers_on_ragged_sizes():
    C.check_dilated_ragged("cuda", cases=((28, 50, 37, 29, 2), (16, 20, 23, 41, 1), (52, 50, 131, 70, 3, 1.0)))


def test_predictor():
    C.check_predict(True)


def test_config1_loss_trajectory(tmp_path):
    """BASELINE.json configs[0]: Segmentor U-Net nb_classes=3 on 8x(256x256), 10 training cycles — the
    reference's CPU run (golden) vs this build on the GPU: train and test loss trajectories."""
    import atomai_amd as aoi
    g = np.load(os.path.join(GOLD, "seg_config1_losses.npz"))
    rs = np.random.RandomState(0)
    X = rs.rand(8, 256, 256).astype(np.float32)
    y = rs.randint(0, 3, (8, 256, 256))
    Xt = rs.rand(8, 256, 256).astype(np.float32)
    yt = rs.randint(0, 3, (8, 256, 256))
    m = aoi.models.Segmentor(nb_classes=3)
    m.fit(X, y, Xt, yt, training_cycles=10, batch_size=8, plot_training_history=False,
          filename=str(tmp_path / "model"))
    assert list(m.batch_idx_train) == list(g["batch_idx_train"])
    # The first steps must agree to the north-star tolerance.  Later ones are compared loosely: Adam's
    # m/sqrt(v) turns rounding-level gradient differences into +-lr parameter moves, so two correct fp32
    # implementations drift apart (the reference's own fp32-vs-fp64 runs do: SURVEY.md §7).
    np.testing.assert_allclose(m.loss_acc["train_loss"][:3], g["train_loss"][:3], rtol=C.REL_TOL)
    np.testing.assert_allclose(m.loss_acc["train_loss"], g["train_loss"], rtol=1e-3)
    np.testing.assert_allclose(m.loss_acc["test_loss"], g["test_loss"], rtol=5e-3)
    assert abs(m.loss_acc["train_loss"][0] - 1.22274) < 1e-4 and abs(m.loss_acc["train_loss"][-1] - 1.11411) < 1e-3


def _oracle_on(device, sd, x, y, ncls, model="Unet", **kw):
    """The oracle's functional network evaluated with torch ops on `device` (full-size checker)."""
    from oracle import seg_oracle as so
    sd = OrderedDict((k, v.to(device)) for k, v in sd.items())
    return so.loss_and_grads(model, sd, x.to(device), y.to(device), ncls, **kw)


@pytest.mark.parametrize("model,ncls,kw", [("SegResNet", 3, dict(batch_norm=False)),
                                           ("SegResNet", 1, dict(layers=[1, 3, 1]))])
def test_variants_vs_oracle(model, ncls, kw):
    C.check_vs_oracle_small(model, ncls, "cuda", **kw)


@pytest.mark.parametrize("model,ncls,B,H", [("Unet", 3, 4, 512), ("dilnet", 1, 2, 256), ("SegResNet", 3, 4, 256),
                                            ("ResHedNet", 3, 2, 128)])
def test_full_width_vs_oracle_on_device(model, ncls, B, H):
    """Default-width nets (nb_filters 16 / 25) at BASELINE resolution: logits, loss and gradients against
    the oracle graph executed with stock torch ops on the same GPU (fp32), errors normalised globally."""
    import atomai_amd as aoi
    torch.manual_seed(1)
    net, _ = aoi.nets.init_fcnn_model(model, ncls)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.rand(B, 1, H, H).astype(np.float32))
    if ncls > 1:
        y = torch.from_numpy(rs.randint(0, ncls, (B, H, H)))
    else:
        y = torch.from_numpy((rs.rand(B, 1, H, H) > 0.5).astype(np.float32))
    net.cuda().train()
    crit = aoi.losses_metrics.select_loss("ce", ncls)
    logits = net(x.cuda())
    loss = crit(logits, y.cuda())
    loss.backward()
    ref_loss, ref_logits, ref_grads = _oracle_on("cuda", sd, x, y, ncls, model)
    # logits and gradients are judged against the oracle in fp64, relative to the noise the oracle's own fp32 run shows
    # against fp64 (deep nets: 12 residual blocks amplify fp32 rounding; SURVEY section 7 "Parity budget"): two correct
    # fp32 implementations differ from each other by up to twice that floor
    from oracle import seg_oracle as so
    y64 = y if ncls > 1 else y.double()
    sd64 = so.cast(sd, torch.float64)
    _, logits64, g64 = _oracle_on("cuda", sd64, x.double(), y64, ncls, model)
    l64 = logits64.cpu().numpy()
    lfloor = C.relmax(ref_logits.cpu().double().numpy(), l64)
    assert C.relmax(logits.detach().cpu().double().numpy(), l64) < max(2 * lfloor, C.REL_TOL), lfloor
    assert abs(loss.item() - float(ref_loss)) / abs(float(ref_loss)) < 1e-5
    # Kink-aware gradient bound (the old `max(4 x floor, 1e-3)` slack was never quantified).
    # LeakyReLU is not differentiable at 0: an input within rounding distance of 0 may take either branch in two correct
    # fp32 implementations, which changes that element's derivative by a factor 100.  The oracle is run once more in fp64
    # with EVERY LeakyReLU input closer than 1e-5 to 0 on the opposite branch (seg_oracle.KINK_FLIP); the gradient change
    # `sens` bounds, to first order, what any choice of branches at those elements can do.  Measured on the MI355X
    # (profiles/r04_fullsize_parity_probe.log): U-Net 512^2 — 1547 such inputs, worst error 4.1e-5 against a reference-fp32
    # floor of 1.2e-4 and sens 1.1e-3: the U-Net is held to the north-star 1e-4 with NO kink or floor allowance; dilnet
    # 2.6e-4 (floor 1.2e-4, sens 5.7e-3), SegResNet 1.3e-4 (= its floor), ResHedNet 1.9e-2 (floor 1.3e-2: 30 layers).
    so.KINK_FLIP = [1e-5, 0]
    try:
        _, _, gflip = _oracle_on("cuda", sd64, x.double(), y64, ncls, model)
        nkink = so.KINK_FLIP[1]
    finally:
        so.KINK_FLIP = None
    gmax = max(float(g.abs().max()) for g in g64.values())
    report = []
    for k, p in net.named_parameters():
        err = float((p.grad.double() - g64[k]).abs().max()) / gmax
        floor = float((ref_grads[k].double() - g64[k]).abs().max()) / gmax
        sens = float((gflip[k] - g64[k]).abs().max()) / gmax
        bound = C.REL_TOL if model == "Unet" else C.REL_TOL + sens + 2 * floor
        report.append((err / bound, k, err, floor, sens))
        assert err < bound, (k, err, floor, sens, nkink)
    worst = max(report)
    print(f"{model}: {nkink} LeakyReLU inputs within 1e-5 of 0; worst gradient error {worst[2]:.2e} ({worst[1]}; reference-fp32 "
          f"floor {worst[3]:.2e}, kink sensitivity {worst[4]:.2e})")


@pytest.mark.parametrize("model,ncls,B,H,seed0", [("dilnet", 1, 1, 32, 71), ("dilnet", 1, 1, 32, 77),
                                                  ("ResHedNet", 3, 1, 16, 113), ("ResHedNet", 3, 1, 16, 200)])
def test_full_width_kink_free_vs_oracle(model, ncls, B, H, seed0):
    """the full-size dilnet / ResHedNet gradient tests pass through a kink allowance (`sens`); is the
    excess over the reference's fp32 floor really LeakyReLU branch flips, or an accumulation problem of the lattice / K-padded
    weight gradients?  Default-WIDTH nets (25 / 50 and 64 / 128 / 256 filters) on an input whose every LeakyReLU
    pre-activation is at least 2e-5 away from 0: seeds are drawn from `seed0` on until the oracle's forward pass ON THIS
    DEVICE says so (`seg_oracle.min_abs_preactivation`; the smallest pre-activation of a deep net moves by ~1e-5 between
    hosts, so the seed is found where the test runs — a frame is small enough for such a draw to exist: the full-size
    frames hold ~7 pre-activations per million inside 1e-5 of the kink).  No branch can flip, so NO `sens` term: every
    gradient within max(1e-4, 2 x the floor the oracle's own fp32 run shows against fp64), normalised globally as in the
    full-size test.  Measured (profiles/r06_fullsize_parity_probe.log): dilnet 7.6e-7 against a floor of 6.5e-7 — the
    full-size excess is branch flips, not accumulation."""
    import atomai_amd as aoi
    from oracle import seg_oracle as so
    for seed in range(seed0, seed0 + 400):
        torch.manual_seed(seed)
        net, _ = aoi.nets.init_fcnn_model(model, ncls)
        sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
        rs = np.random.RandomState(seed)
        x = torch.from_numpy(rs.rand(B, 1, H, H).astype(np.float32))
        y = (torch.from_numpy(rs.randint(0, ncls, (B, H, H))) if ncls > 1
             else torch.from_numpy((rs.rand(B, 1, H, H) > 0.5).astype(np.float32)))
        sdc = OrderedDict((k, v.cuda()) for k, v in sd.items())
        if so.min_abs_preactivation(model, sdc, x.cuda()) > 2e-5 and \
                so.min_abs_preactivation(model, so.cast(sdc, torch.float64), x.double().cuda()) > 2e-5:
            break
    else:
        pytest.fail("no kink-free draw in 400 seeds")
    net.cuda().train()
    loss = aoi.losses_metrics.select_loss("ce", ncls)(net(x.cuda()), y.cuda())
    loss.backward()
    y64 = y if ncls > 1 else y.double()
    l32, _, g32 = _oracle_on("cuda", sd, x, y, ncls, model)
    l64, _, g64 = _oracle_on("cuda", so.cast(sd, torch.float64), x.double(), y64, ncls, model)
    assert abs(loss.item() - float(l64)) / abs(float(l64)) < 1e-5
    gmax = max(float(g.abs().max()) for g in g64.values())
    report = []
    for k, p in net.named_parameters():
        err = float((p.grad.double() - g64[k]).abs().max()) / gmax
        floor = float((g32[k].double() - g64[k]).abs().max()) / gmax
        # (ResHedNet: 12 residual blocks amplify fp32 rounding until single tensors sit at 6e-5 of the global scale in the
        #  ORACLE's own fp32 run; one realisation of that noise is a rough estimate of its size — 3 x there, 2 x for dilnet)
        bound = max(C.REL_TOL, (3 if model == "ResHedNet" else 2) * floor)
        report.append((err / bound, k, err, floor))
        assert err < bound, (k, err, floor)
    worst = max(report)
    print(f"{model} kink-free {B}x{H}^2 seed {seed}: worst gradient error {worst[2]:.2e} ({worst[1]}; reference-fp32 floor "
          f"{worst[3]:.2e}); largest err / floor {max(r[2] / max(r[3], 1e-12) for r in report):.2f}")


@pytest.mark.parametrize("C0,Co,dil", [(25, 50, 2), (50, 50, 4), (50, 50, 6), (50, 25, 1), (50, 50, 2)])
def test_dilnet_layer_gradients_are_linear_exact_at_full_size(C0, Co, dil):
    """The dilated (lattice-mode) and remainder-column weight / data gradients at dilnet's config-3 layer shapes
    (512^2 pooled frames, 28 / 52 stored channels, split-K over 256 workgroups) WITHOUT an activation: conv -> BatchNorm
    is differentiable everywhere, so the error against the fp64 torch graph is pure accumulation error.  Each tensor
    within max(1e-5, 2 x the torch-fp32 floor) of its own largest entry — a 2x regression of the split-K reduction or of
    the K-padded lattice weight gradient fails here."""
    import torch.nn.functional as F
    import torch.nn as nn
    from atomai_amd.engine import Tape
    torch.manual_seed(dil * 100 + C0)
    B, H = 2, 512
    conv = nn.Conv2d(C0, Co, 3, padding=dil, dilation=dil).cuda()
    bn = nn.BatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    x = torch.randn(B, C0, H, H, device="cuda", requires_grad=True)
    tape = Tape(True, True)
    n = tape.input(x)
    o = tape.output(tape.conv([n.out], conv, bn, 1.0))
    gy = torch.randn_like(o.value)
    o.grad_out = gy
    tape.backward()
    got = [n.grad_nchw] + [tape.param_grads[id(p)][1] for p in (conv.weight, conv.bias, bn.weight, bn.bias)]

    def torch_graph(dtype):
        xr = x.detach().to(dtype).requires_grad_(True)
        ps = [p.detach().to(dtype).requires_grad_(True) for p in (conv.weight, conv.bias, bn.weight, bn.bias)]
        yy = F.batch_norm(F.conv2d(xr, ps[0], ps[1], padding=dil, dilation=dil), None, None, ps[2], ps[3], True)
        return yy.detach(), torch.autograd.grad(yy, [xr] + ps, gy.to(dtype))
    y64, g64 = torch_graph(torch.float64)
    y32, g32 = torch_graph(torch.float32)
    assert C.relmax(o.value.cpu().double().numpy(), y64.cpu().numpy()) < max(1e-5, 2 * C.relmax(y32.cpu().double().numpy(), y64.cpu().numpy()))
    report = []
    for nm, a, r64, r32 in zip(("dx", "dW", "db", "dgamma", "dbeta"), got, g64, g32):
        # (db of conv -> BatchNorm is exactly 0 in exact arithmetic: judged on the scale of dbeta)
        sc = max(float(r64.abs().max()), 1e-3 * float(g64[4].abs().max()))
        err = float((a.view_as(r64).double() - r64).abs().max()) / sc
        floor = float((r32.double() - r64).abs().max()) / sc
        report.append((err / max(1e-5, 2 * floor), nm, err, floor))
        assert err < max(1e-5, 2 * floor), (nm, err, floor)
    worst = max(report)
    print(f"dilnet layer {C0}->{Co} dilation {dil} @ {H}^2 x {B}, no activation: worst {worst[1]} error {worst[2]:.2e} "
          f"(torch-fp32 floor {worst[3]:.2e}); dW error {report[1][2]:.2e} (floor {report[1][3]:.2e})")


def test_config2_three_step_trajectory_vs_oracle_on_device():
    """BASELINE configs[1] at its FULL size — Segmentor U-Net nb_classes=3, 512x512, bs 32, three Adam steps — against
    oracle.seg_oracle.train_step (trainer.py:189-211 restated) executed with stock torch ops on the device (VERDICT r03
    weak #2: 'nothing compares a bs-32, 512^2, k-step loss trajectory with the oracle').  Measured (profiles/
    r04_fullsize_parity_probe.log): 5.6e-9 / 7.6e-7 / 2.2e-6 relative to the fp64 oracle — the oracle's own fp32 run is at
    1.1e-6 / 7.6e-7 / 1.6e-6; asserted: 1e-5 on step 1, the north-star 1e-4 on steps 2 and 3 (Adam's m / sqrt(v) turns
    rounding-level gradient differences into +-lr parameter moves, so later steps drift)."""
    import atomai_amd as aoi
    from oracle import seg_oracle as so
    rs = np.random.RandomState(0)
    X = rs.rand(32, 512, 512).astype(np.float32)
    y = rs.randint(0, 3, (32, 512, 512))
    m = aoi.models.Segmentor(nb_classes=3, seed=1)
    m.compile_trainer((X, y, X, y), training_cycles=3, batch_size=32)
    sd = OrderedDict((k, v.detach().clone().cuda()) for k, v in m.net.state_dict().items())
    xb, yb = m.X_train[0].detach().clone().cuda(), m.y_train[0].detach().clone().cuda()
    opt = so.AdamState(lr=1e-3)
    got, ref = [], []
    for step in range(3):
        got.append(m.train_step(m.X_train[0], m.y_train[0])[0])
        ref.append(so.train_step("Unet", sd, opt, xb, yb, 3))
        if step == 0:
            # BatchNorm running statistics after the first step: the same batch statistics on both sides (measured
            # 4.5e-7).  Not compared later: Adam's first update is -lr * sign(g), so parameters whose gradient is at
            # rounding level move in opposite directions in two correct fp32 runs (2e-3 apart after one step, measured),
            # and the running statistics follow them (6e-4 / 2.4e-3 after steps 2 / 3) while the loss stays within 2e-6.
            for k, v in m.net.state_dict().items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    assert float((v.double() - sd[k].double()).abs().max()) < 1e-5 * max(1.0, float(sd[k].abs().max())), k
    rel = [abs(a - b) / abs(b) for a, b in zip(got, ref)]
    assert rel[0] < 1e-5 and max(rel) < C.REL_TOL, (got, ref)
    assert got[2] < got[1] < got[0]


def test_determinism_and_loss_decrease_at_full_size():
    """Size-independent properties at the BASELINE configuration (bs=32, 512x512): two identical runs are
    bit-identical (no float atomics anywhere) and the loss goes down."""
    import atomai_amd as aoi
    rs = np.random.RandomState(0)
    X = rs.rand(32, 512, 512).astype(np.float32)
    y = rs.randint(0, 3, (32, 512, 512))
    outs = []
    for _ in range(2):
        m = aoi.models.Segmentor(nb_classes=3, seed=1)
        m.compile_trainer((X, y, X, y), training_cycles=4, batch_size=32)
        ls = [m.train_step(m.X_train[0], m.y_train[0])[0] for _ in range(4)]
        outs.append((ls, [p.detach().clone() for p in m.net.parameters()]))
    assert outs[0][0] == outs[1][0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    assert outs[0][0][-1] < outs[0][0][0]


def test_kernel_level_large_shapes():
    """wgrad / dgrad kernels at config-2 layer shapes against torch's conv -> LeakyReLU -> BatchNorm backward on the same
    device, in the kink-aware form of the full-size net tests (this test used a relative-L2 bound of
    1e-3).  Reference = the torch graph in fp64; floor = the same graph in fp32 against it; sens = the fp64 gradient
    change when every LeakyReLU input within 1e-5 of 0 takes the other branch (what two correct fp32 implementations may
    legitimately disagree on).  Per tensor, max-abs normalised by that tensor's largest entry:
    err < 1e-4 + sens + 2 x floor.  Measured: profiles/r05_fullsize_parity_probe.log (the printed lines)."""
    import torch.nn.functional as F
    from atomai_amd.engine import Tape
    import torch.nn as nn
    torch.manual_seed(0)
    for (B, H, C0, C1, Co) in [(8, 128, 64, 64, 64), (4, 256, 32, 0, 32), (2, 512, 16, 16, 16), (8, 64, 128, 0, 128)]:
        conv = nn.Conv2d(C0 + C1, Co, 3, padding=1).cuda()
        bn = nn.BatchNorm2d(Co).cuda()
        xs = [torch.randn(B, c, H, H, device="cuda", requires_grad=True) for c in (C0, C1) if c]
        tape = Tape(True, True)
        ins = [tape.input(x) for x in xs]
        out = tape.conv([n.out for n in ins], conv, bn, 0.01)
        o = tape.output(out)
        gy = torch.randn_like(o.value)
        o.grad_out = gy
        tape.backward()
        got = [n.grad_nchw for n in ins] + [tape.param_grads[id(p)][1] for p in
                                            (conv.weight, conv.bias, bn.weight, bn.bias)]
        names = [f"dx{i}" for i in range(len(xs))] + ["dW", "db", "dgamma", "dbeta"]

        def torch_graph(dtype, flip=None):
            xr = [x.detach().to(dtype).requires_grad_(True) for x in xs]
            ps = [p.detach().to(dtype).requires_grad_(True) for p in (conv.weight, conv.bias, bn.weight, bn.bias)]
            pre = F.conv2d(torch.cat(xr, 1), ps[0], ps[1], padding=1)
            nk = 0
            if flip is None:
                act = F.leaky_relu(pre, 0.01)
            else:
                near = pre.detach().abs() < flip
                nk = int(near.sum())
                neg = (pre.detach() < 0) ^ near
                act = torch.where(neg, 0.01 * pre, pre)
            y = F.batch_norm(act, None, None, ps[2], ps[3], True)
            return y.detach(), torch.autograd.grad(y, xr + ps, gy.to(dtype)), nk
        y64, g64, _ = torch_graph(torch.float64)
        y32, g32, _ = torch_graph(torch.float32)
        _, gflip, nkink = torch_graph(torch.float64, flip=1e-5)
        yfloor = C.relmax(y32.cpu().double().numpy(), y64.cpu().numpy())
        assert C.relmax(o.value.cpu().double().numpy(), y64.cpu().numpy()) < max(C.REL_TOL, 2 * yfloor)
        report = []
        for nm, a, r64, r32, rf in zip(names, got, g64, g32, gflip):
            sc = float(r64.abs().max())
            err = float((a.view_as(r64).double() - r64).abs().max()) / sc
            floor = float((r32.double() - r64).abs().max()) / sc
            sens = float((rf - r64).abs().max()) / sc
            bound = C.REL_TOL + sens + 2 * floor
            report.append((err / bound, nm, err, floor, sens))
            assert err < bound, (B, H, C0, C1, Co, nm, err, floor, sens, nkink)
        worst = max(report)
        print(f"conv block {C0}+{C1}->{Co} @ {H}^2 x {B}: {nkink} LeakyReLU inputs within 1e-5 of 0; worst {worst[1]} error "
              f"{worst[2]:.2e} (torch-fp32 floor {worst[3]:.2e}, kink sensitivity {worst[4]:.2e})")


def test_config3_dilnet_predict_full_size_vs_oracle_on_device():
    """BASELINE.json configs[2] at full frame size: default dilnet (nb_filters 25), 1024x1024 frames, through
    SegPredictor's chunked pipeline (several chunks in flight, device-side normalisation, kernel download) against the
    oracle's eval-mode graph executed with stock torch ops on the same GPU; plus chunk-size independence (bit-exact)."""
    import atomai_amd as aoi
    from oracle import seg_oracle as so
    torch.manual_seed(3)
    net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
    with torch.no_grad():                                    # non-trivial BatchNorm running statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    rs = np.random.RandomState(5)
    stack = rs.rand(5, 1024, 1024).astype(np.float32)
    p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False,
                                    chunk_bytes=8 << 20)      # 2 frames per chunk -> 3 chunks, last one ragged
    out = p.run(stack, compute_coords=False)
    assert out.shape == (5, 1024, 1024, 1) and p._norm is not None
    x = (stack - stack.min()) / np.ptp(stack)                 # the reference's host-side normalisation
    ref = so.predict_probs("dilnet", OrderedDict((k, v.cuda()) for k, v in sd.items()),
                           torch.from_numpy(x[:, None]).cuda(), 1).cpu().numpy()
    assert C.relmax(out, ref.astype(np.float64)) < C.REL_TOL
    p.chunk_bytes = 64 << 20                                  # one chunk
    assert np.array_equal(p.run(stack, compute_coords=False), out)


def test_config3_dilnet_predict_64_frames_properties():
    """BASELINE.json configs[2] on a 64-frame 1024x1024 stack (the full 4096 frames are the same pipeline repeated):
    size-independent properties — frames are independent in eval mode (a shuffled stack gives the shuffled output, bit
    for bit, also across chunk boundaries), the result does not depend on the chunking, every probability lies in
    [0, 1], and the first frames equal the oracle's eval graph under the GLOBAL min-max normalisation of the stack."""
    import atomai_amd as aoi
    from oracle import seg_oracle as so
    torch.manual_seed(3)
    net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    rs = np.random.RandomState(6)
    stack = rs.rand(64, 1024, 1024).astype(np.float32)
    stack[17] *= 1.5                                          # the global maximum lives in one frame
    p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
    out = p.run(stack, compute_coords=False)
    assert out.shape == (64, 1024, 1024, 1)
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0 and np.isfinite(out).all()
    perm = rs.permutation(64)
    assert np.array_equal(p.run(stack[perm], compute_coords=False), out[perm])
    p.chunk_bytes = 20 << 20                                  # 5 frames per chunk: ragged last chunk
    assert np.array_equal(p.run(stack, compute_coords=False), out)
    x = ((stack[:3] - stack.min()) / np.ptp(stack))[:, None]
    ref = so.predict_probs("dilnet", OrderedDict((k, v.cuda()) for k, v in sd.items()), torch.from_numpy(x).cuda(), 1)
    assert C.relmax(out[:3], ref.cpu().numpy().astype(np.float64)) < C.REL_TOL


@pytest.mark.parametrize("bn", [True, False])
def test_convblock_training_dropout(bn):
    import _dropout_checks as D
    D.check_convblock_dropout("cuda", batch_norm=bn)
    D.check_convblock_dropout("cuda", N=3, Cin=16, Cout=40, H=70, W=50, p=0.5, batch_norm=bn)


def test_dilatedblock_dropout_vs_reference_golden():
    import _dropout_checks as D
    D.check_dilated_dropout_golden("cuda")


def test_dilatedblock_without_batchnorm_gradients():
    import _dropout_checks as D
    D.check_dilated_no_batchnorm("cuda")


def test_input_normalisation_inside_the_first_layer_kernel():
    C.check_input_norm_fusion("cuda")


def test_classification_head_in_the_last_conv_epilogue():
    C.check_head_fusion("cuda")


def test_dilated_block_sum_in_the_last_conv_epilogue():
    C.check_dsum_fusion("cuda")


def test_iou_vs_reference_golden():
    import _metrics_checks as M
    M.check_iou_golden("cuda")


def test_iou_many_classes_and_probabilities_in():
    import _metrics_checks as M
    M.check_iou_wide_golden("cuda")


def test_predict_with_more_than_eight_classes():
    C.check_many_classes_predict("cuda")


def test_fit_with_compute_accuracy(tmp_path):
    import _metrics_checks as M
    M.check_fit_with_accuracy(True, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 16), (16, 16)])
def test_wave_specialised_thin_conv(cin, cout, monkeypatch):
    """conv_ws.hip against the general kernel and fp64 autograd (128x128 images: 2 tiles per CU and more)."""
    C.check_wave_specialised_conv("cuda", cin, cout, monkeypatch, hw=128, batch=12)


@pytest.mark.gpu
def test_wave_specialised_two_source_layer(monkeypatch):
    C.check_wave_specialised_concat("cuda", monkeypatch, hw=128, batch=12)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,H,N", [(32, 16, 200, 6), (16, 32, 131, 5), (32, 32, 128, 8), (64, 32, 77, 7),
                                          (32, 64, 96, 4), (64, 64, 64, 9), (128, 128, 40, 8), (64, 16, 90, 3)])
def test_wave_specialised_wgrad_is_bit_identical(cin, cout, H, N, monkeypatch):
    """wgrad_ws.hip (producer / consumer waves) against wgrad_kernel.h: many tiles per persistent workgroup, ragged
    image sides; partial rows bit-identical, their sum against fp64 autograd."""
    C.check_wgrad_ws_bit_identical("cuda", cin, cout, H, N, monkeypatch)
    if (cin, cout) == (16, 32):
        C.check_wgrad_ws_bit_identical("cuda", cin, cout, H, N, monkeypatch, force_th=8)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 16), (16, 16)])
def test_bn_backward_formed_in_the_loaders(cin, cout, monkeypatch):
    """amx_conv2d_dgrad_fused / amx_conv2d_wgrad_fused against amx_bn_bwd_apply + the plain kernels (bit-identical)."""
    C.check_bwd_fused_in_loaders("cuda", cin, cout, monkeypatch, hw=128, batch=12)


@pytest.mark.gpu
def test_bn_backward_formed_in_the_loaders_unet(monkeypatch):
    assert C.check_bwd_fused_in_loaders("cuda", 0, 0, monkeypatch, hw=256, batch=16, unet=True) == 4


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(8, 16), (32, 32)])
def test_bn_backward_formed_in_the_loaders_resblock(cin, cout, monkeypatch):
    """ResBlocks (16 / 32 channels, >= 2 x CUs tiles) through the fused loaders; 6 repetitions must agree bit
    for bit with each other (a side-stream / main-stream race on the shared gradient tensor would not) and with
    AMX_BWD_FUSE=0."""
    assert C.check_bwd_fused_in_loaders("cuda", cin, cout, monkeypatch, hw=128, batch=12, res=True, repeats=6) >= 2


@pytest.mark.gpu
def test_loss_upstream_gradient_factor():
    C.check_loss_upstream_gradient("cuda")


@pytest.mark.gpu
def test_eval_pool_inside_the_first_layer_kernel():
    C.check_pool_fusion("cuda")


@pytest.mark.gpu
def test_upsample_forward_is_exact():
    C.check_upsample_exact("cuda")


@pytest.mark.gpu
def test_remainder_column_classes_vs_padded_plan():
    """25 / 50-filter layers on the 16 + 3 x 4 / 3 x 16 + 4 column plan (v_mfma_f32_4x4x1 remainder blocks)."""
    C.check_remainder_columns("cuda")


def test_upsample_block_as_one_launch_inside_the_nets():
    C.check_upconv_node("cuda", hw=64, batch=4)


def test_upsample_block_forward_in_one_pass_is_bit_identical():
    C.check_upconv_fused_kernel("cuda")
    C.check_upconv_fused_kernel("cuda", cases=((50, 25, 3, 512, 512, 0), (32, 16, 8, 256, 256, 0), (128, 64, 4, 64, 64, 1)))


def test_hooked_block_by_block_forward_equals_fused():
    C.check_hooked_forward_equals_fused("cuda")


def test_lattice_xpack_is_bit_identical(monkeypatch):
    C.check_lattice_xpack_bit_identical("cuda", monkeypatch)


def test_two_source_data_gradient_as_two_wave_specialised_launches(monkeypatch):
    # (the wave-specialised kernel takes a launch from two 16 x 16 tiles per CU on: 2 x 256^2 = 512 tiles on the MI355X)
    C.check_split_two_source_dgrad("cuda", monkeypatch, hw=256, batch=2)


def test_head_and_loss_of_the_training_step_in_one_pass():
    C.check_fused_head_and_loss("cuda")


def test_pool_backward_fused_with_the_first_layer_weight_gradient():
    C.check_pool_backward_with_first_layer_wgrad("cuda")
