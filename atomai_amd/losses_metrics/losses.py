"""Segmentation losses on the HIP path (reference: atomai/losses_metrics/losses.py:139-174).

``select_loss('ce', nb_classes)`` returns a module with the same call signature as
``torch.nn.CrossEntropyLoss()`` / ``torch.nn.BCEWithLogitsLoss()`` (mean reduction), whose forward is ONE
kernel that also produces d loss / d logits, so ``loss.backward()`` costs nothing extra on this op.
"""
import torch
import torch.nn as nn

import os

from .. import _lib as L

# workgroups of the fused loss kernels (each walks its pixels with a grid stride, one dependent load chain per thread):
# 1024 blocks left 4 waves per SIMD to hide a ~2 us load round trip (2.2 TB/s on the 267 MB of a bs-32 512^2 CE pass)
LOSS_BLOCKS = [int(os.environ.get("AMX_LOSS_BLOCKS", "4096"))]


def _times_upstream(dl, g):
    """dlogits * (upstream gradient of the scalar loss) without a pass over dlogits when the upstream gradient is 1
    (``loss.backward()``): the factor is read on the device, nothing synchronises."""
    if g.numel() == 1 and g.dtype == torch.float32 and g.device == dl.device:
        L.call("amx_scale_unless_one", L.ptr(dl), L.ptr(g.contiguous()), dl.numel(), L.stream_ptr(dl))
        return dl
    return dl * g


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        N, K = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        x = logits.detach().contiguous()
        t = target.contiguous()
        need = logits.requires_grad
        dl = torch.empty_like(x) if need else None
        rows = max(1, min(LOSS_BLOCKS[0], (N * HW + 255) // 256))
        part = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.call("amx_ce_fwd_bwd", L.ptr(x), L.ptr(t), L.ptr(dl), L.ptr(part), rows, N, K, HW,
               L.stream_ptr(x))
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.call("amx_reduce_rows", L.ptr(part), rows, 1, 1, 1.0 / (N * HW), L.ptr(loss), L.stream_ptr(x))
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, ctx.dl = ctx.dl, None
        return _times_upstream(dl, g), None


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        x = logits.detach().contiguous()
        t = target.detach().contiguous().to(torch.float32)
        n = x.numel()
        need = logits.requires_grad
        dl = torch.empty_like(x) if need else None
        rows = max(1, min(LOSS_BLOCKS[0], (n + 255) // 256))
        part = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.call("amx_bce_fwd_bwd", L.ptr(x), L.ptr(t), L.ptr(dl), L.ptr(part), rows, n, L.stream_ptr(x))
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.call("amx_reduce_rows", L.ptr(part), rows, 1, 1, 1.0 / n, L.ptr(loss), L.stream_ptr(x))
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, ctx.dl = ctx.dl, None
        return _times_upstream(dl, g), None


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss() semantics: logits (N,K,H,W), int64 targets (N,H,W), mean."""

    def forward(self, logits, target):
        if target.dtype != torch.int64 or logits.ndim < 3 or target.shape != logits.shape[:1] + logits.shape[2:]:
            raise ValueError("expected logits (N,K,...) and int64 targets (N,...)")
        return _CEFn.apply(logits, target)

    def __repr__(self):
        return "CrossEntropyLoss()"


class BCEWithLogitsLoss(nn.Module):
    """torch.nn.BCEWithLogitsLoss() semantics (mean over all elements)."""

    def forward(self, logits, target):
        if target.shape != logits.shape:
            raise ValueError(f"Target size ({target.shape}) must be the same as input size ({logits.shape})")
        return _BCEFn.apply(logits, target)

    def __repr__(self):
        return "BCEWithLogitsLoss()"


def select_loss(loss: str, nb_classes: int = None, **kwargs):
    """Same selection logic and error behaviour as the reference (losses.py:139-174) for the losses on
    the hot path ('ce', callables); the others are outside this build's scope."""
    if loss in ['ce', 'multitask'] and nb_classes is None:
        raise ValueError("For cross-entropy loss function, you must specify the number of classes")
    if loss == 'ce' and nb_classes == 1:
        return BCEWithLogitsLoss()
    if loss == 'ce' and nb_classes > 2:
        return CrossEntropyLoss()
    if loss == 'mse':
        return torch.nn.MSELoss()
    if hasattr(loss, "__call__"):
        return loss
    if loss in ('dice', 'focal', 'nll', 'multitask_nll', 'multitask_ce'):
        raise NotImplementedError(f"loss '{loss}' is outside the MI355X hot path of this build")
    raise NotImplementedError(
        "Select Dice loss ('dice'), focal loss ('focal') "
        " cross-entropy loss ('ce'), means-squared error ('mse'),"
        " multitask loss (multitask_nll and multitask_ce)"
        " or pass your custom loss function")
