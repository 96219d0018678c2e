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
