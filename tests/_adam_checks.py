"""Shared body of the isolated Adam parity test (emulator tier / gpu tier): amx_adam_flat + optim.FusedAdam against
torch.optim.Adam in float64 on IDENTICAL injected gradients (reference: atomai/trainers/trainer.py:539 lr 1e-3,
vitrainer.py:218 lr 1e-4 — torch.optim.Adam defaults)."""
import numpy as np
import torch


def check_adam_flat_kernel(device, n=1003, steps=10, lr=1e-3, gscale=0.25):
    """The raw entry point: n NOT a multiple of 4 (scalar tail), gscale != 1, 10 steps (bias-correction exponents)."""
    from atomai_amd import _lib as L
    rs = np.random.RandomState(7)
    p0 = rs.randn(n).astype(np.float32)
    grads = [rs.randn(n).astype(np.float32) * (10.0 ** rs.uniform(-4, 1)) for _ in range(steps)]
    npad = (n + 3) // 4 * 4
    bufs = {k: torch.zeros(npad, dtype=torch.float32, device=device) for k in ("p", "g", "m", "v")}
    bufs["p"][:n] = torch.from_numpy(p0).to(device)
    ref_p = torch.nn.Parameter(torch.from_numpy(p0).double())
    ref = torch.optim.Adam([ref_p], lr=lr)
    b1, b2, eps = 0.9, 0.999, 1e-8
    for t, g in enumerate(grads, 1):
        bufs["g"][:n] = torch.from_numpy(g).to(device)
        L.call("amx_adam_flat", L.ptr(bufs["p"]), L.ptr(bufs["g"]), L.ptr(bufs["m"]), L.ptr(bufs["v"]), n, lr, b1, b2,
               eps, 1.0 - b1 ** t, 1.0 - b2 ** t, gscale, L.stream_ptr(bufs["p"]))
        ref_p.grad = torch.from_numpy(g).double() * gscale
        ref.step()
        st = ref.state[ref_p]
        for name, got, want in (("p", bufs["p"], ref_p.detach()), ("m", bufs["m"], st["exp_avg"]),
                                ("v", bufs["v"], st["exp_avg_sq"])):
            got = got[:n].cpu().double().numpy()
            want = want.numpy()
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
            assert err < 1e-6, f"step {t} {name}: rel err {err:.3e}"
    # the padding slots beyond n must not have been touched by the scalar tail
    assert float(bufs["p"][n:].abs().max()) == 0.0 if npad > n else True


def check_fused_adam_vs_torch(device, steps=10):
    """The optimizer object on several odd-sized parameters, against torch.optim.Adam(float64); also the
    'parameter without a gradient is skipped' rule and the state_dict format."""
    from atomai_amd.optim import FusedAdam
    rs = np.random.RandomState(3)
    shapes = [(5, 3, 3, 3), (5,), (7, 5, 1, 1), (1,), (13, 2)]
    ps = [torch.nn.Parameter(torch.from_numpy(rs.randn(*s).astype(np.float32)).to(device)) for s in shapes]
    ref_ps = [torch.nn.Parameter(p.detach().cpu().double().clone()) for p in ps]
    opt = FusedAdam(ps, lr=1e-4)
    opt.prepare()
    ref = torch.optim.Adam(ref_ps, lr=1e-4)
    for t in range(steps):
        opt.zero_grad()
        ref.zero_grad()
        for i, (p, q) in enumerate(zip(ps, ref_ps)):
            if t == 4 and i == 2:
                continue                                   # no gradient for this parameter on this step: skipped
            g = rs.randn(*p.shape).astype(np.float32)
            p.grad = torch.from_numpy(g).to(device)
            q.grad = torch.from_numpy(g).double()
        opt.step()
        ref.step()
        for i, (p, q) in enumerate(zip(ps, ref_ps)):
            err = float((p.detach().cpu().double() - q.detach()).abs().max() / q.detach().abs().max())
            assert err < 1e-6, f"step {t} param {i}: {err:.3e}"
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][2]["step"]) == steps - 1 and float(sd["state"][0]["step"]) == steps
