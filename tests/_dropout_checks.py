"""Shared body of the training-mode Dropout tests (ConvBlock: Conv2d -> Dropout -> LeakyReLU -> BatchNorm2d)."""
import copy

import numpy as np
import torch


def check_convblock_dropout(device, N=2, Cin=3, Cout=8, H=12, W=10, p=0.3, batch_norm=True):
    """Injected masks: forward, input gradient and every parameter gradient against the stock-torch fp64 graph
    conv -> mask -> LeakyReLU -> BatchNorm; then the in-kernel generator: drop fraction, scaling, determinism under
    torch.manual_seed, eval mode = identity."""
    import torch.nn as nn
    import atomai_amd.engine as eng
    from atomai_amd.nets import ConvBlock
    torch.manual_seed(0)
    m = ConvBlock(2, 2, Cin, Cout, batch_norm=batch_norm, dropout_=p)
    ref_layers = [copy.deepcopy(l).double() for l in m.block]
    m.to(device).train()
    rs = np.random.RandomState(0)
    masks = [torch.from_numpy((rs.rand(N, Cout, H, W) >= p).astype(np.float32) / (1 - p)) for _ in range(2)]
    calls = []

    def hook(shape, pp):
        assert abs(pp - p) < 1e-12
        mk = masks[len(calls)]
        calls.append(shape)
        out = torch.zeros(shape)
        out[..., :Cout] = mk.permute(0, 2, 3, 1)
        return out
    x = torch.from_numpy(rs.randn(N, Cin, H, W).astype(np.float32))
    gy = torch.from_numpy(rs.randn(N, Cout, H, W).astype(np.float32))
    x1 = x.clone().to(device).requires_grad_(True)
    eng.DROPOUT_MASK_HOOK[0] = hook
    try:
        y = m(x1)
        y.backward(gy.to(device))
    finally:
        eng.DROPOUT_MASK_HOOK[0] = None
    assert len(calls) == 2
    # reference graph
    x2 = x.double().requires_grad_(True)
    h, li = x2, 0
    for l in ref_layers:
        if isinstance(l, nn.Dropout):
            h = h * masks[li].double()
            li += 1
        else:
            l.train()
            h = l(h)
    h.backward(gy.double())
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max())   # noqa: E731
    assert rel(y, h) < 1e-4
    assert rel(x1.grad, x2.grad) < 1e-4
    ref_params = [q for l in ref_layers for q in l.parameters()]
    for (k, q), r in zip(m.block.named_parameters(), ref_params):
        assert rel(q.grad, r.grad) < 1e-4, k
    if batch_norm:
        bn_ref = [l for l in ref_layers if isinstance(l, nn.BatchNorm2d)][0]
        bn_ours = [l for l in m.block if isinstance(l, nn.BatchNorm2d)][0]
        assert rel(bn_ours.running_var, bn_ref.running_var) < 1e-4
    # ---- the generator
    big = ConvBlock(2, 1, 1, 16, batch_norm=False, dropout_=0.25).to(device).train()
    xb = torch.ones(2, 1, 64, 64, device=device)
    torch.manual_seed(5)
    a = big(xb).detach().cpu()
    torch.manual_seed(5)
    b = big(xb).detach().cpu()
    c = big(xb).detach().cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)
    big.eval()
    e = big(xb).detach().cpu()
    zero = (a == 0).float().mean().item()
    assert abs(zero - 0.25) < 0.01
    keep = a != 0
    assert float(((a[keep] - e[keep] / 0.75).abs() / e[keep].abs().clamp_min(1e-6)).max()) < 1e-5


def check_dilated_dropout_golden(device):
    """DilatedBlock with Dropout layers vs tests/golden/dilated_dropout.npz (the reference's own module graph with the
    random masks made explicit — oracle/make_golden.py dil_drop): training forward (every sub-layer output summed, the
    Dropout layer's included), input and parameter gradients, BatchNorm running statistics, eval forward (Dropout is
    the identity there, so the convolution output counts twice)."""
    import os
    import atomai_amd.engine as eng
    from atomai_amd.nets.blocks import DilatedBlock
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dilated_dropout.npz"))
    for tag in ("bn", "nobn"):
        meta = [int(v) for v in g[f"{tag}|meta"]]
        bn, cin, cout, H, W, p100, dils = bool(meta[0]), meta[1], meta[2], meta[3], meta[4], meta[5], meta[6:]
        blk = DilatedBlock(2, cin, cout, dils, dils, batch_norm=bn, dropout_=p100 / 100)
        blk.load_state_dict({k: torch.from_numpy(g[f"{tag}|w|{k}"]) for k in blk.state_dict()})
        blk.to(device).train()
        masks = g[f"{tag}|masks"]
        calls = []

        def hook(shape, pp):
            mk = torch.from_numpy(masks[len(calls)])
            calls.append(shape)
            out = torch.zeros(shape)
            out[..., :cout] = mk.permute(0, 2, 3, 1)
            return out
        x = torch.from_numpy(g[f"{tag}|x"]).to(device).requires_grad_(True)
        eng.DROPOUT_MASK_HOOK[0] = hook
        try:
            y = blk(x)
            y.backward(torch.from_numpy(g[f"{tag}|gy"]).to(device))
        finally:
            eng.DROPOUT_MASK_HOOK[0] = None
        assert len(calls) == len(dils)

        def close(got, key, what):
            r64, r32 = g[f"{tag}|{key}|f64"], g[f"{tag}|{key}|f32"]
            floor = np.abs(r32.astype(np.float64) - r64).max()
            err = np.abs(got.detach().cpu().double().numpy() - r64).max()
            assert err <= max(4 * floor, 1e-4 * np.abs(r64).max()), (tag, what, err, floor)
        close(y, "y", "train forward")
        close(x.grad, "dx", "input gradient")
        for k, q in blk.named_parameters():
            close(q.grad, f"grad|{k}", k)
        for k, v in blk.state_dict().items():
            if "running" in k:
                close(v, f"bn|{k}", k)
        blk.eval()
        with torch.no_grad():
            close(blk(torch.from_numpy(g[f"{tag}|x"]).to(device)), "y_eval", "eval forward")


def check_dilated_no_batchnorm(device):
    """DilatedBlock(batch_norm=False): the block sums (conv output, activation) per layer; forward and every gradient
    against the stock-torch fp64 graph of the same modules (the no-BatchNorm backward double-counted the shared sum
    gradient until round 3)."""
    from atomai_amd.nets.blocks import DilatedBlock
    for dils, cin, cout, H, W in (([2, 4], 3, 6, 12, 10), ([2, 4, 6], 8, 20, 19, 23)):
        torch.manual_seed(0)
        m = DilatedBlock(2, cin, cout, dils, dils, batch_norm=False)
        ref = [copy.deepcopy(l).double() for l in m.atrous_module]
        m.to(device).train()
        x = torch.randn(2, cin, H, W)
        gy = torch.randn(2, cout, H, W)
        x1 = x.clone().to(device).requires_grad_(True)
        y = m(x1)
        y.backward(gy.to(device))
        x2 = x.double().requires_grad_(True)
        h, outs = x2, []
        for l in ref:
            h = l(h)
            outs.append(h)
        yr = sum(outs)
        yr.backward(gy.double())
        rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max())   # noqa: E731
        assert rel(y, yr) < 1e-5
        assert rel(x1.grad, x2.grad) < 1e-4
        for (k, q), r in zip(m.atrous_module.named_parameters(), [q for l in ref for q in l.parameters()]):
            assert rel(q.grad, r.grad) < 1e-4, k
