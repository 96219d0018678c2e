"""Dense layers on the hand-written fp32-MFMA GEMM (csrc/linear.hip): ``linear(x, weight, bias, act)`` ==
``act(F.linear(x, weight, bias))`` with forward, data gradient and weight gradient all on ``amx_gemm_f32``.

The owning modules keep their ``nn.Linear`` children as parameter containers (state-dict keys, initialisation order and
checkpoints stay the reference's: atomai/nets/ed.py:292-343, 530-580; atomai/nets/gp.py:14-26) and call this function in
``forward``.  fp64 inputs (dklGPR's precision='double') go to the library GEMM: the fp32 matrix cores cannot serve them.
"""
import torch
import torch.nn.functional as F

from .. import _lib as L

ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


def _gemm(A, sam, sak, B, sbk, sbn, M, N, K, bias=None, act=0, out=None):
    """C[M][N] = act(A * B + bias); ``out``: write into this (M, N) contiguous tensor (a view of the optimizer's flat
    gradient bucket for weight gradients) instead of a fresh one.  Long-K problems with few output tiles are split
    along k (deterministic two-stage sum, csrc/linear.hip)."""
    C = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    splits = L.load().amx_gemm_f32_splits(M, N, K)
    if splits > 1:
        work = torch.empty(splits * M * N, dtype=torch.float32, device=A.device)
        L.call("amx_gemm_f32_splitk", L.ptr(A), sam, sak, L.ptr(B), sbk, sbn, L.ptr(C), N, L.ptr(bias), M, N, K, act,
               L.ptr(work), splits, L.stream_ptr(A))
    else:
        L.call("amx_gemm_f32", L.ptr(A), sam, sak, L.ptr(B), sbk, sbn, L.ptr(C), N, L.ptr(bias), M, N, K, act,
               L.stream_ptr(A))
    return C


def _grad_target(param, shape, like):
    """The optimizer's flat-bucket view for this parameter's gradient (engine.grad_buffer) when it can be written
    directly — the FusedAdam step then finds the gradient in place instead of copying it (17 device-to-device copies
    per rVAE step, 80 us) — else None."""
    from ..engine import grad_buffer
    if param is None or getattr(param, "_amx_grad", None) is None or tuple(param.shape) != tuple(shape):
        return None
    v = grad_buffer(param, like)
    return v if (v.data_ptr() == param._amx_grad.data_ptr() and v.is_contiguous()) else None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous()
        w = weight.detach().contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = _gemm(x2, K, 1, w, 1, K, M, N, K, None if bias is None else bias.detach().contiguous(), act)
        ctx.act, ctx.has_bias = act, bias is not None
        ctx.wparam, ctx.bparam = weight, bias                # (the Parameters themselves: their flat-bucket gradient views)
        ctx.save_for_backward(x2, w, y if act else None)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dpre = dy.reshape(M, N).contiguous()
        sp = L.stream_ptr(dpre)
        if ctx.act:
            t = torch.empty_like(dpre)
            L.call("amx_act_bwd", L.ptr(dpre), L.ptr(y), L.ptr(t), dpre.numel(), ctx.act, sp)
            dpre = t
        dx = dw = db = None
        if ctx.needs_input_grad[0]:                      # dx[M][K] = dpre[M][N] * W[N][K]
            dx = _gemm(dpre, N, 1, w, K, 1, M, K, N).reshape(*dy.shape[:-1], K)
        if ctx.needs_input_grad[1]:                      # dW[N][K] = dpre^T[N][M] * x[M][K]
            tgt = _grad_target(ctx.wparam, (N, K), dpre)
            dw = _gemm(dpre, 1, N, x2, K, 1, N, K, M, out=tgt)
            if tgt is not None:
                dw = dw.view(N, K)                       # a fresh tensor object over the bucket: autograd adopts it as .grad
        if ctx.has_bias and ctx.needs_input_grad[2]:     # column sums in (at most) two deterministic stages
            rows, src = M, dpre
            if rows > 64:
                nch = 32
                tmp = torch.empty(nch, N, dtype=torch.float32, device=dpre.device)
                L.call("amx_reduce_rows_chunked", L.ptr(src), rows, N, nch, L.ptr(tmp), sp)
                src, rows = tmp, -(-rows // -(-rows // nch))
            tgt = _grad_target(ctx.bparam, (N,), dpre)
            db = tgt.view(N) if tgt is not None else torch.empty(N, dtype=torch.float32, device=dpre.device)
            L.call("amx_reduce_rows_chunked", L.ptr(src), rows, N, 1, L.ptr(db), sp)
        ctx.wparam = ctx.bparam = None
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None, act=None) -> torch.Tensor:
    a = ACT[act]
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        y = F.linear(x, weight, bias)                    # fp64 (DKL precision='double'): library GEMM
        return torch.tanh(y) if a == 1 else (torch.relu(y) if a == 2 else y)
    return _LinearFn.apply(x, weight, bias, a)


def run_dense(seq, x: torch.Tensor) -> torch.Tensor:
    """An nn.Sequential of Linear / Tanh / ReLU modules evaluated with every activation fused into the GEMM before it."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            act = "tanh" if isinstance(nxt, torch.nn.Tanh) else ("relu" if isinstance(nxt, torch.nn.ReLU) else None)
            x = linear(x, m.weight, m.bias, act)
            i += 2 if act else 1
        else:
            x = m(x)
            i += 1
    return x
