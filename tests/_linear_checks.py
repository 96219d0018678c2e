"""Shared bodies of the dense-layer (fp32 MFMA GEMM, csrc/linear.hip) tests: emulator tier / gpu tier."""
import numpy as np
import torch


def check_gemm_strides(device, sizes=((37, 5, 4099), (64, 64, 16), (130, 70, 33), (1, 1, 1), (5, 300, 129))):
    """amx_gemm_f32 in its three operand layouts (x W^T, dpre W, dpre^T x), odd sizes, unaligned bases, every
    activation — against float64 matmul."""
    from atomai_amd import _lib as L
    rs = np.random.RandomState(0)
    for (M, N, K) in sizes:
        x = torch.from_numpy(rs.randn(M, K).astype(np.float32)).to(device)
        w = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).to(device)
        b = torch.from_numpy(rs.randn(N).astype(np.float32)).to(device)
        sp = L.stream_ptr(x)
        for act, fn in ((0, lambda t: t), (1, torch.tanh), (2, torch.relu)):
            y = torch.empty(M, N, device=device)
            L.call("amx_gemm_f32", L.ptr(x), K, 1, L.ptr(w), 1, K, L.ptr(y), N, L.ptr(b), M, N, K, act, sp)
            ref = fn(x.double().cpu() @ w.double().cpu().T + b.double().cpu())
            assert float((y.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (M, N, K, act)
        d = torch.from_numpy(rs.randn(M, N).astype(np.float32)).to(device)
        dx = torch.empty(M, K, device=device)                      # dx = d W        (B contiguous along n)
        L.call("amx_gemm_f32", L.ptr(d), N, 1, L.ptr(w), K, 1, L.ptr(dx), K, None, M, K, N, 0, sp)
        ref = d.double().cpu() @ w.double().cpu()
        assert float((dx.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        dw = torch.empty(N, K, device=device)                      # dW = d^T x      (A contiguous along m)
        L.call("amx_gemm_f32", L.ptr(d), 1, N, L.ptr(x), K, 1, L.ptr(dw), K, None, N, K, M, 0, sp)
        ref = d.double().cpu().T @ x.double().cpu()
        assert float((dw.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # an operand whose base pointer is not 16-byte aligned takes the scalar loader
    big = torch.from_numpy(rs.randn(9 * 16 + 1).astype(np.float32)).to(device)
    xo = big[1:].view(9, 16)
    w = torch.from_numpy(rs.randn(3, 16).astype(np.float32)).to(device)
    y = torch.empty(9, 3, device=device)
    L.call("amx_gemm_f32", L.ptr(xo), 16, 1, L.ptr(w), 1, 16, L.ptr(y), 3, None, 9, 3, 16, 0, L.stream_ptr(y))
    assert float((y.cpu().double() - xo.double().cpu() @ w.double().cpu().T).abs().max()) < 1e-5


def check_linear_autograd(device):
    """nets._linear.linear / run_dense: value and all three gradients == torch.nn.functional.linear (+ act) in fp64,
    for a 3-D input and every activation; bit-identical run to run."""
    from atomai_amd.nets._linear import linear, run_dense
    rs = np.random.RandomState(1)
    for act, fn in (("tanh", torch.tanh), ("relu", torch.relu), (None, lambda t: t)):
        x = torch.from_numpy(rs.randn(3, 70, 45).astype(np.float32)).to(device).requires_grad_(True)
        w = torch.from_numpy((rs.randn(21, 45) / 6).astype(np.float32)).to(device).requires_grad_(True)
        b = torch.from_numpy(rs.randn(21).astype(np.float32)).to(device).requires_grad_(True)
        g = torch.from_numpy(rs.randn(3, 70, 21).astype(np.float32)).to(device)
        y = linear(x, w, b, act)
        y.backward(g)
        xr, wr, br = (t.detach().cpu().double().requires_grad_(True) for t in (x, w, b))
        yr = fn(torch.nn.functional.linear(xr, wr, br))
        yr.backward(g.cpu().double())
        for a_, r_ in ((y, yr), (x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
            assert float((a_.detach().cpu().double() - r_.detach()).abs().max() / r_.detach().abs().max()) < 1e-5
        y2 = linear(x.detach(), w.detach(), b.detach(), act)
        assert torch.equal(y2, y.detach())
    seq = torch.nn.Sequential(torch.nn.Linear(12, 9), torch.nn.ReLU(), torch.nn.Linear(9, 4), torch.nn.Tanh(),
                              torch.nn.Linear(4, 2)).to(device)
    x = torch.from_numpy(rs.randn(11, 12).astype(np.float32)).to(device)
    ref = seq.double()(x.double()) if device == "cpu" else seq(x).double()
    seq.float()
    assert float((run_dense(seq, x).double() - ref).abs().max()) < 1e-5


def check_gemm_splitk(device, sizes=((37, 5, 4099), (40, 24, 1500), (512, 128, 4096))):
    """Long-K problems with few output tiles go through the split-K path (amx_gemm_f32_splitk: k slices summed in slice
    order, then bias / activation): against float64 matmul, bit-identical between two runs, and reached through
    nets/_linear.linear with its autograd."""
    from atomai_amd import _lib as L
    from atomai_amd.nets._linear import linear
    rs = np.random.RandomState(1)
    for (M, N, K) in sizes:
        splits = L.load().amx_gemm_f32_splits(M, N, K)
        assert splits > 1, (M, N, K)
        x = torch.from_numpy(rs.randn(M, K).astype(np.float32)).to(device)
        w = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).to(device)
        b = torch.from_numpy(rs.randn(N).astype(np.float32)).to(device)
        ref = torch.tanh(x.double().cpu() @ w.double().cpu().T + b.double().cpu())
        outs = []
        for _ in range(2):
            y = torch.empty(M, N, device=device)
            work = torch.empty(splits * M * N, device=device)
            L.call("amx_gemm_f32_splitk", L.ptr(x), K, 1, L.ptr(w), 1, K, L.ptr(y), N, L.ptr(b), M, N, K, 1, L.ptr(work),
                   splits, L.stream_ptr(x))
            outs.append(y.cpu())
        assert torch.equal(outs[0], outs[1])
        assert float((outs[0].double() - ref).abs().max()) < 2e-5
        wp = w.clone().requires_grad_(True)
        yl = linear(x, wp, b, "tanh")
        assert float((yl.detach().cpu().double() - ref).abs().max()) < 2e-5
        yl.sum().backward()
        xr, wr = x.double().cpu(), w.double().cpu().requires_grad_(True)
        torch.tanh(xr @ wr.T + b.double().cpu()).sum().backward()
        assert float((wp.grad.cpu().double() - wr.grad).abs().max()) < 1e-4 * max(1.0, float(wr.grad.abs().max()))
    assert L.load().amx_gemm_f32_splits(16384, 1000, 256) == 1 and L.load().amx_gemm_f32_splits(64, 64, 512) == 1
