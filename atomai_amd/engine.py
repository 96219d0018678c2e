"""Define-by-run tape over the HIP kernels (include/atomai_amd.h): forward AND hand-written backward.

Why not torch.autograd per op: the fused pipeline passes *lazy* activations between layers — the raw
post-LeakyReLU tensor plus the producer's BatchNorm (scale, shift), which the consumer applies while
loading (conv_fwd.hip).  Gradients are therefore defined w.r.t. the *normalised* value while the tensor
that exists in HBM is the raw one; a hand-written backward keeps that bookkeeping explicit and lets
dgrad / wgrad / BN-backward share buffers.  The whole tape is exposed to PyTorch as ONE
autograd.Function (nets/_function.py), so ``loss.backward()`` / ``optimizer.step()`` work unchanged
(atomai/trainers/trainer.py:201-207).

Everything here is plumbing: shapes, workspaces (torch allocator), launch order.  All arithmetic is in
the kernels.  No CPU fallback exists; `_lib` raises if the extension is missing.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

import os as _os
# Backward fusion switches, chosen by measurement:
#   2: BN-backward sums emitted by the pool / px backward kernels that write the final dy   (default: on)
#   4: dpre formed in the wgrad loader (experiment; off)
# Interleaved in-process A/B on MI355X in round 1 (U-Net bs 32 512^2, ms/step, min of 3):
#   0: 20.76   2: 20.22   [dgrad-loader / dgrad-epilogue fusions, bits 1 and 8: 21.22 - 22.40]
# -> moving the HBM-bound BatchNorm-backward passes INTO the MFMA kernels costs them more (registers ->
#    fewer co-resident workgroups, exposed load latency) than the removed passes save.  Bits 1 and 8 (fusion into the
#    data-gradient kernel) were removed in round 2 together with their register cost in the convolution kernel.
FUSE = int(_os.environ.get("AMX_FUSE", "2")) & 6
# first-layer weight gradient on the main stream (host-side switch, read once; tools/gpu_step_ab.py re-reads it for A/B)
FIRST_WGRAD_MAIN = _os.environ.get("AMX_FIRST_WGRAD_MAIN", "1") != "0"
# data gradient of a two-source layer as one wave-specialised launch per source where the kernel takes them singly (host-side)
DGRAD_SPLIT = _os.environ.get("AMX_DGRAD_SPLIT", "1") != "0"
# pooling backward fused with the first layer's weight gradient (amx_pool2x2_bwd_wgrad1): dy of that layer never reaches HBM
FUSE_POOL_WGRAD1 = _os.environ.get("AMX_FUSE_POOL_WGRAD1", "1") != "0"


def bwd_fuse_enabled() -> bool:
    """AMX_BWD_FUSE (csrc/knobs.hip, default 1): BatchNorm / LeakyReLU backward formed inside the loaders of the
    wave-specialised data- and weight-gradient kernels.  The library and the host read the same resolved switch."""
    return L.knob("AMX_BWD_FUSE") != 0


def r4(c: int) -> int:
    return (c + 3) // 4 * 4


def r16(c: int) -> int:
    return (c + 15) // 16 * 16


# TEST HOOK: callable(shape, p) -> mask tensor (values 0 or 1/(1-p)) used instead of the in-kernel generator
DROPOUT_MASK_HOOK = [None]

# generation counter bumped by the fused optimizer (it writes weights through raw pointers, which does
# not touch torch's version counters); part of the packed-weight cache key
_weight_generation = [0]


def bump_weight_generation() -> None:
    _weight_generation[0] += 1


class Act:
    """An activation as it lives in HBM: NHWC fp32 [N,H,W,Cs] (Cs = C padded to 4) + the pending
    per-channel affine of the producing BatchNorm (None == identity)."""
    __slots__ = ("t", "N", "H", "W", "C", "Cs", "scale", "shift", "grad", "gx", "needs_grad", "producer",
                 "first_consumer", "bstats", "post_slope", "unmasked", "pooled", "wg1", "wg1_parts")

    def __init__(self, t, C, scale=None, shift=None, needs_grad=False):
        self.t = t
        self.N, self.H, self.W, self.Cs = t.shape
        self.C = C
        self.scale, self.shift = scale, shift
        self.post_slope = 1.0                         # LeakyReLU applied by the consumer AFTER the pending affine
                                                      # (ResBlock's conv -> BN -> activation order); 1.0 = none
        self.grad: Optional[torch.Tensor] = None      # d loss / d (value the consumers see), NHWC
        self.gx: Optional[torch.Tensor] = None        # DilatedBlock's shared extra gradient
        self.unmasked: Optional[torch.Tensor] = None  # DilatedBlock + training Dropout: lrelu(conv) BEFORE the mask
        self.pooled: Optional["Act"] = None           # eval mode: max_pool2d of this (normalised) activation, already computed
        self.wg1 = None                               # output of the net's FIRST layer (training): (net input [N,1,H,W], slope)
                                                      # -> the pooling backward may form that layer's weight-gradient sums
        self.wg1_parts = None                         # ... and leaves them here: (part3 [rows][3][10][Cs], rows)
        self.needs_grad = needs_grad
        # No object references from an activation back to graph nodes: node <-> activation cycles would keep
        # gigabytes of device memory alive until the cyclic garbage collector happens to run (erratic step times).
        self.producer = False                         # True: written by a ConvNode that owns a pending BatchNorm
        self.first_consumer = None                    # id() of the first node that read this activation in forward
                                                      # order == the LAST one to add to `grad` in backward order
        self.bstats = None                            # (rows tensor, rows, stride, column offset): per-block
                                                      # (sum dy, sum dy*a) emitted by that last writer

    def consumed_by(self, node) -> None:
        if self.first_consumer is None:
            self.first_consumer = id(node)            # nodes stay alive in tape.nodes for as long as this is used

    def wants_bstats(self, node, training: bool) -> bool:
        """True if `node` writes the final gradient of this activation and its producer needs BN-backward sums."""
        if isinstance(node, ConvNode):
            return False                                  # the data-gradient kernel emits no statistics
        return (bool(FUSE & 2) and training and self.producer and self.gx is None
                and self.needs_grad and self.first_consumer == id(node))

    def wants_bsum_from_dgrad(self, node, training: bool) -> bool:
        """True if the data gradient `node` is about to write is the COMPLETE gradient of this activation (sole writer)
        and its producer owns a BatchNorm: the wave-specialised data-gradient kernel can then emit the BatchNorm-backward
        sums of that producer in its epilogue (round 6, amx_conv2d_dgrad_fused_bsum)."""
        return (bool(FUSE & 2) and training and self.producer and self.gx is None and self.grad is None
                and self.needs_grad and self.first_consumer == id(node))

    @property
    def npix(self) -> int:
        return self.N * self.H * self.W


class _PackCache:
    """Packed weight images keyed by (tensor identity, layout), valid for one (storage, version, optimizer
    generation).  Entries hold only a weak reference to the parameter: they die with it, and an `id()` that gets
    reused by a later tensor can never produce a stale hit.

    Weight images (entries with a layout `meta`) are refreshed TOGETHER: the first miss after an optimizer step
    re-packs every image used during the previous generation in place with one batched launch per 8 layers
    (`amx_pack_weights_batch`), instead of 30 small launches scattered between the convolutions of the step."""

    def __init__(self):
        self.store: Dict[tuple, list] = {}          # key -> [ver, value, weakref, meta, last generation used]
        self.refreshed_gen = -1

    @staticmethod
    def _ver(w):
        return (w.data_ptr(), w._version, _weight_generation[0])

    def get(self, w: torch.Tensor, key_extra: tuple, build, meta=None):
        key = (id(w),) + key_extra
        gen = _weight_generation[0]
        hit = self.store.get(key)
        if hit is not None and hit[2]() is w:
            if hit[0] != self._ver(w) and meta is not None and self.refreshed_gen != gen:
                self._refresh_all(w.device)
            if hit[0] == self._ver(w):
                hit[4] = gen
                return hit[1]
        val = build()
        if len(self.store) > 4096:                               # long sessions (ensembles): drop dead entries
            self.store = {k: v for k, v in self.store.items() if v[2]() is not None}
        self.store[key] = [self._ver(w), val, weakref.ref(w), meta, gen]
        return val

    def _refresh_all(self, device) -> None:
        gen = _weight_generation[0]
        self.refreshed_gen = gen
        jobs = []
        for ent in self.store.values():
            w = ent[2]()
            if (w is None or ent[3] is None or ent[4] < gen - 1 or w.device != device
                    or ent[0] == self._ver(w)):
                continue
            jobs.append((ent, w))
        if not jobs:
            return
        n = len(jobs)
        wp = (ctypes.c_void_p * n)(*[w.data_ptr() for _, w in jobs])
        dp = (ctypes.c_void_p * n)(*[ent[1].data_ptr() for ent, _ in jobs])
        desc = (ctypes.c_int * (7 * n))(*[v for ent, w in jobs for v in (w.shape[0],) + tuple(ent[3])])
        L.call("amx_pack_weights_batch", wp, dp, desc, n, _sp(jobs[0][1]))
        for ent, w in jobs:
            ent[0] = self._ver(w)


_pack_cache = _PackCache()

_bn_generation = [0]          # bumped by every training-mode amx_bn_finalize (it updates running stats through raw pointers)
_eval_cache: Dict[tuple, tuple] = {}


def cached_eval(owner, tensors, extra: tuple, build):
    """Eval-mode constants derived from parameters / BatchNorm buffers only (the eval affine, the folded head weights):
    built once per (tensor storage, tensor version, optimizer generation, BatchNorm generation) instead of on every
    predictor chunk."""
    ver = tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (_weight_generation[0], _bn_generation[0])
    key = (id(owner),) + extra
    hit = _eval_cache.get(key)
    if hit is not None and hit[0]() is owner and hit[1] == ver:
        return hit[2]
    val = build()
    if len(_eval_cache) > 4096:
        for k in [k for k, v in _eval_cache.items() if v[0]() is None]:
            del _eval_cache[k]
    _eval_cache[key] = (weakref.ref(owner), ver, val)
    return val


def bn_eval_affine(bn, cout: int, cos: int, like: torch.Tensor):
    """(scale, shift) of an eval-mode BatchNorm, cached."""
    def build():
        scale, shift = _empty((cos,), like), _empty((cos,), like)
        L.call("amx_bn_eval_affine", L.ptr(bn.weight.detach()), L.ptr(bn.bias.detach()), L.ptr(bn.running_mean),
               L.ptr(bn.running_var), bn.eps, cout, cos, L.ptr(scale), L.ptr(shift), _sp(like))
        return scale, shift
    return cached_eval(bn, (bn.weight, bn.bias, bn.running_mean, bn.running_var),
                       ("affine", cout, cos, float(bn.eps), like.device.index), build)


def _empty(shape, like: torch.Tensor, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _sp(t):
    return L.stream_ptr(t)


def grad_buffer(p: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Where a kernel should write d loss / d p: the optimizer's flat-bucket view when the parameter has
    one, no gradient is pending accumulation and the view has not been handed out yet in this generation
    (optim.FusedAdam), else a fresh tensor."""
    v = getattr(p, "_amx_grad", None)
    if (v is not None and p.grad is None and not getattr(p, "_amx_grad_busy", False)
            and v.shape == p.shape and v.device == like.device):
        # handed out at most ONCE per optimizer generation (FusedAdam.zero_grad / step clear the mark): a parameter
        # used twice in one tape, or by two tapes before one backward, must not get two aliases of the same memory
        # (autograd would then sum the last write with itself instead of G1 + G2)
        p._amx_grad_busy = True
        return v
    return torch.empty(p.shape, dtype=torch.float32, device=like.device)


def pack_weights(w: torch.Tensor, C0, C0s, C1, C1s, taps, mode) -> torch.Tensor:
    def build():
        n = L.load().amx_pack_weights_size(w.shape[0], C0s, C1s, taps, mode)
        dst = _empty((n,), w)
        L.call("amx_pack_weights", L.ptr(w.detach()), L.ptr(dst), w.shape[0], C0, C0s, C1, C1s, taps,
               mode, _sp(w))
        return dst
    return _pack_cache.get(w, (C0, C0s, C1, C1s, taps, mode), build, meta=(C0, C0s, C1, C1s, taps, mode))


def pack_weights_range(w: torch.Tensor, ci_off: int, Cn: int, Cns: int, taps: int, mode: int) -> torch.Tensor:
    """Image of the convolution over the input channels [ci_off, ci_off + Cn) of `w` alone (amx_pack_weights_range)."""
    def build():
        n = L.load().amx_pack_weights_size(w.shape[0], Cns, 0, taps, mode)
        dst = _empty((n,), w)
        L.call("amx_pack_weights_range", L.ptr(w.detach()), L.ptr(dst), w.shape[0], w.shape[1], ci_off, Cn, Cns, taps,
               mode, _sp(w))
        return dst
    return _pack_cache.get(w, ("range", ci_off, Cn, Cns, taps, mode), build)


def padded_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    """bias vector zero-padded to n (cached like the packed weights)."""
    def build():
        out = torch.zeros(n, dtype=torch.float32, device=v.device)
        out[: v.numel()].copy_(v.detach())
        return out
    return _pack_cache.get(v, ("pad", n), build)


# ====================================================================================== nodes
class _Node:
    def backward(self, tape: "Tape") -> None:
        raise NotImplementedError


class ConvNode(_Node):
    """conv (3x3 / dilated / 1x1) [+bias] [+LeakyReLU] [+BatchNorm statistics] over 1 or 2 sources."""

    def __init__(self, tape, srcs: Sequence[Act], conv, bn, slope: float, x_plain=None, post_slope: float = 1.0,
                 drop_p: float = 0.0, keep_unmasked: bool = False, pool_next: bool = False):
        self.srcs = list(srcs)
        self.pool_next = pool_next                    # first layer whose output goes straight into a 2x2 max-pool
        self.keep_unmasked = keep_unmasked            # DilatedBlock sums the convolution output itself too
        self.conv, self.bn, self.slope = conv, bn, float(slope)
        # training-mode nn.Dropout between the convolution and its LeakyReLU (blocks.py:68-69): applied after the fused
        # conv + activation (dropout.hip); the mask is kept for backward
        self.drop_p = float(drop_p) if tape.training else 0.0
        self.mask = None
        # activation AFTER the BatchNorm (ResBlock): without a BatchNorm it is simply the epilogue activation
        self.post_slope = float(post_slope)
        if self.post_slope != 1.0 and bn is None:
            assert self.slope == 1.0
            self.slope, self.post_slope = self.post_slope, 1.0
        assert self.post_slope == 1.0 or self.slope == 1.0, "one activation per layer"
        self.x_plain = x_plain                           # (N,1,H,W) tensor for the Cin==1 kernel
        w = conv.weight
        self.cout = w.shape[0]
        self.taps = w.shape[2] * w.shape[3]
        assert self.taps in (1, 9) and w.shape[2] == w.shape[3], "3x3 or 1x1 kernels only"
        self.dil = int(conv.dilation[0]) if self.taps == 9 else 1
        assert tuple(conv.stride) == (1, 1)
        if self.taps == 9:
            assert int(conv.padding[0]) == self.dil and int(conv.padding[1]) == self.dil, \
                "padding must equal dilation ('same' convolution)"
        else:
            assert tuple(conv.padding) == (0, 0)
        for src in self.srcs:
            src.consumed_by(self)
        self.out = self._forward(tape)
        self.out.post_slope = self.post_slope
        self.out.producer = self.bn is not None and self.post_slope == 1.0
        # the net's first layer in training (1 input channel, 3x3, BatchNorm, no dropout mask): its consumer's pooling
        # backward may fuse this layer's weight gradient (PoolNode.backward, amx_pool2x2_bwd_wgrad1)
        if (FUSE_POOL_WGRAD1 and self.x_plain is not None and self.bn is not None and tape.training and tape.need_grad
                and self.mask is None and self.post_slope == 1.0 and self.taps == 9 and self.x_plain.shape[1] == 1
                and bwd_fuse_enabled() and bool(FUSE & 2) and not (FUSE & 4)):      # (FUSE & 4: the experiment mode that
            self.out.wg1 = (self.x_plain, float(self.slope))                            #  materialises dpre from the real dy)

    # -------------------------------------------------------------------------------- forward
    def _forward(self, tape) -> Act:
        w, b = self.conv.weight, self.conv.bias
        cos, cop = r4(self.cout), r16(self.cout)
        training_bn = self.bn is not None and tape.training
        drop = self.drop_p > 0.0
        want_stats = training_bn
        if drop:
            training_bn = False                              # the statistics are those of the MASKED tensor (below)
        if self.x_plain is not None:
            x = self.x_plain
            N, _, H, W = x.shape
            y = _empty((N, H, W, cos), x)
            npix = N * H * W
            self.rows = L.load().amx_rows_for(npix)
            self.rows_pix = L.load().amx_rows_pix(npix)
            stats = _empty((self.rows, 2, cop), x) if training_bn else None
            norm = tape.input_norm if tape.input_norm is not None else (0.0, 1.0)
            tape.input_norm_used = tape.input_norm is not None
            self.pooled = None
            if (self.pool_next and FUSE_POOL and self.bn is not None and not tape.training and not tape.need_grad
                    and not drop and L.load().amx_conv1_fwd_pool_supported(H, W, self.dil, self.rows_pix)):
                # eval mode: the 2x2 max-pool of the normalised output comes out of the same kernel (fcnn.py:123, 219)
                psc, psh = bn_eval_affine(self.bn, self.cout, cos, y)
                self.pooled = _empty((N, H // 2, W // 2, cos), x)
                L.call("amx_conv1_fwd_pool", L.ptr(x), L.ptr(w.detach()), L.ptr(b.detach() if b is not None else None),
                       L.ptr(y), L.ptr(self.pooled), L.ptr(psc), L.ptr(psh), N, H, W, self.cout, cos, self.dil,
                       self.slope, self.rows, self.rows_pix, float(norm[0]), float(norm[1]), _sp(x))
            else:
                L.call("amx_conv1_fwd", L.ptr(x), L.ptr(w.detach()), L.ptr(b.detach() if b is not None else None),
                       L.ptr(y), L.ptr(stats), N, H, W, self.cout, cos, self.dil, self.slope, self.rows,
                       self.rows_pix, float(norm[0]), float(norm[1]), _sp(x))
            stat_mode, lat = 1, 0
        else:
            s0 = self.srcs[0]
            s1 = self.srcs[1] if len(self.srcs) > 1 else None
            N, H, W = s0.N, s0.H, s0.W
            C0, C0s = s0.C, s0.Cs
            C1, C1s = (s1.C, s1.Cs) if s1 else (0, 0)
            assert w.shape[1] == C0 + C1, "channel mismatch between conv weight and its sources"
            wpk = pack_weights(w, C0, C0s, C1, C1s, self.taps, 0)
            bias = b.detach() if b is not None else None
            y = _empty((N, H, W, cos), s0.t)
            th = L.load().amx_conv2d_tile_h(C0s + C1s, self.cout, self.taps, self.dil, H)
            self.rows = L.load().amx_conv2d_stats_rows(C0s + C1s, self.cout, self.taps, self.dil, N, H, W)
            self.rows_pix = th                               # modes 0 / 3: the rows one wave owns
            lat = L.load().amx_conv2d_stats_lattice(self.taps, self.dil)
            stats = _empty((self.rows, 2, cop), s0.t) if training_bn else None
            ps0, ps1 = s0.post_slope, (s1.post_slope if s1 else 1.0)
            if ps0 != 1.0 or ps1 != 1.0:
                L.call("amx_conv2d_fwd_act", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), ps0, C0s,
                       L.ptr(s1.t if s1 else None), L.ptr(s1.scale if s1 else None),
                       L.ptr(s1.shift if s1 else None), ps1, C1s, L.ptr(wpk), L.ptr(bias), None,
                       L.ptr(y), cos, None, 0, L.ptr(stats), N, H, W, self.cout, self.taps, self.dil,
                       self.slope, _sp(y))
            else:
                L.call("amx_conv2d_fwd", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), C0s,
                       L.ptr(s1.t if s1 else None), L.ptr(s1.scale if s1 else None),
                       L.ptr(s1.shift if s1 else None), C1s, L.ptr(wpk), L.ptr(bias), None,
                       L.ptr(y), cos, None, 0, L.ptr(stats), N, H, W, self.cout, self.taps, self.dil,
                       self.slope, _sp(y))
            stat_mode = 3 if lat else 0                      # 3: rows ordered by residue class (dilated layers)
        if drop:
            npix = N * H * W
            self.rows = L.load().amx_rows_for(npix)
            self.rows_pix = L.load().amx_rows_pix(npix)
            stats = _empty((self.rows, 2, cop), y) if want_stats else None
            self.mask = _empty(y.shape, y)
            unmasked = y.clone() if self.keep_unmasked else None
            hook = DROPOUT_MASK_HOOK[0]
            mask_in = hook(tuple(y.shape), self.drop_p).to(y.device).float().contiguous() if hook else None
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # CPU generator: reproducible under manual_seed
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed ^= torch.distributed.get_rank() << 48             # data-parallel ranks draw different masks
            L.call("amx_dropout_fwd", L.ptr(y), L.ptr(self.mask), L.ptr(mask_in), self.drop_p, seed, L.ptr(stats),
                   npix, cos, cop, self.rows, self.rows_pix, _sp(y))
            stat_mode = 1
            training_bn = want_stats
        scale = shift = None
        if self.bn is not None:
            bn = self.bn
            if tape.training:
                scale, shift = _empty((cos,), y), _empty((cos,), y)
                _bn_generation[0] += 1
                self.save_mean, self.save_invstd = _empty((cos,), y), _empty((cos,), y)
                mom = BN_MOMENTUM if bn.momentum is None else bn.momentum
                nrows = self.rows
                if nrows > 512 or stat_mode == 3:                  # two-stage merge: coalesced chunk merge, then per channel
                    nch = max(1, min(1024, nrows // 32))           # (the only consumer of the lattice row order)
                    merged = _empty((nch, 3, cop), y)
                    L.call("amx_bn_stats_merge", L.ptr(stats), nrows, cop, stat_mode, N, H, W,
                           self.rows_pix, lat if stat_mode == 3 else 1, nch, L.ptr(merged), _sp(y))
                    stats, nrows, stat_mode = merged, -(-nrows // -(-nrows // nch)), 2
                L.call("amx_bn_finalize", L.ptr(stats), nrows, cop, stat_mode, N, H, W, self.rows_pix,
                       L.ptr(bn.weight.detach()), L.ptr(bn.bias.detach()), L.ptr(bn.running_mean),
                       L.ptr(bn.running_var), mom, bn.eps, self.cout, cos, L.ptr(scale), L.ptr(shift),
                       L.ptr(self.save_mean), L.ptr(self.save_invstd), _sp(y))
                if bn.num_batches_tracked is not None:
                    tape.bn_counters.append(bn.num_batches_tracked)      # += 1 for all layers in ONE launch
            else:
                scale, shift = bn_eval_affine(bn, self.cout, cos, y)
        needs = tape.need_grad
        act = Act(y, self.cout, scale, shift, needs_grad=needs)
        if self.x_plain is not None and self.pooled is not None:
            act.pooled = Act(self.pooled, self.cout, needs_grad=False)
        if drop and self.keep_unmasked:
            act.unmasked = unmasked
        return act

    # -------------------------------------------------------------------------------- backward
    def backward(self, tape) -> None:
        """BatchNorm backward: per-channel sums (emitted by the pool / px backward kernel that wrote the final dy, or
        by amx_bn_bwd_reduce) -> amx_bn_bwd_finalize -> dpre = lrelu'(a) * (k1*dy + k2*a + k3) materialised by
        amx_bn_bwd_apply; then the weight gradient (side stream) and the data gradient read dpre."""
        out = self.out
        dy = out.grad if out.grad is not None else out.gx
        if dy is None:
            return                                        # output never used downstream
        a = out.t
        npix, cos = out.npix, out.Cs
        sp = _sp(a)
        has_bias = self.conv.bias is not None
        if self.bn is not None and not tape.training:
            raise L.AmxError("backward through eval-mode BatchNorm is not on the hot path")
        if self.post_slope != 1.0:
            # consumers saw LeakyReLU(scale*a + shift): first back through that activation, then the usual
            # BatchNorm backward (with an identity epilogue activation, self.slope == 1)
            assert out.gx is None
            masked = _empty(a.shape, a)
            L.call("amx_lrelu_bwd", L.ptr(dy), L.ptr(a), L.ptr(out.scale), L.ptr(out.shift), self.post_slope,
                   npix, cos, L.ptr(masked), None, sp)
            dy = masked
        fused = out.gx is None and bool(FUSE & 4) and self.mask is None
        # Round 4: layers whose data gradient runs on the wave-specialised kernel form dpre inside BOTH consumers' loaders
        # (amx_conv2d_dgrad_fused / amx_conv2d_wgrad_fused) — no amx_bn_bwd_apply pass, dpre never exists in HBM.
        fused_ws = (not fused and self._bwd_fusable(out)) if self.bn is not None else False
        # the net's FIRST layer has no data gradient: its BatchNorm backward is only read by the first-layer weight-gradient
        # kernel, which forms it while loading (amx_conv1_wgrad_fused) — the 512^2 x 16-channel apply pass of U-Net's c1
        # (0.32 ms at the very end of the backward pass, nothing left to overlap it) is not launched at all
        if (not fused and self.bn is not None and self.x_plain is not None and self.mask is None and out.gx is None
                and self.post_slope == 1.0 and bwd_fuse_enabled()):
            fused_ws = True
        fused = fused or fused_ws
        k = None
        aux = None
        bias_part = None
        if self.bn is not None:
            bn = self.bn
            if out.bstats is not None:
                part, rows, stride, off = out.bstats
                part_ptr = part.view(-1)[off:]
            else:
                rows = L.load().amx_rows_for(npix)
                part = _empty((rows, 2, cos), a)
                L.call("amx_bn_bwd_reduce", L.ptr(dy), L.ptr(a), npix, cos, L.ptr(part), sp)
                part_ptr, stride = part, cos
            dgamma, dbeta = grad_buffer(bn.weight, a), grad_buffer(bn.bias, a)
            k = _empty((3, cos), a)
            L.call("amx_bn_bwd_finalize", L.ptr(part_ptr), rows, stride, cos, self.cout, npix,
                   L.ptr(bn.weight.detach()), L.ptr(self.save_mean), L.ptr(self.save_invstd),
                   L.ptr(dgamma), L.ptr(dbeta), L.ptr(k[0]), L.ptr(k[1]), L.ptr(k[2]), sp)
            tape.add_param_grad(bn.weight, dgamma)
            tape.add_param_grad(bn.bias, dbeta)
        needs_transform = self.bn is not None or self.slope != 1.0 or out.gx is not None
        dpre_mat = None
        if needs_transform and fused and not fused_ws:
            # experiment mode: one of the two consumers still wants a materialised dpre
            arows = L.load().amx_rows_for(npix)
            bias_part = _empty((arows, cos), a) if has_bias else None
            dpre_mat = _empty(a.shape, a)
            kk = (L.ptr(k[0]), L.ptr(k[1]), L.ptr(k[2])) if k is not None else (None, None, None)
            L.call("amx_bn_bwd_apply", L.ptr(dy), L.ptr(a), L.ptr(out.gx), *kk, self.slope, npix, cos,
                   L.ptr(dpre_mat), L.ptr(bias_part), sp)
        if needs_transform and fused:
            aux, dpre = a, dy                               # transformed on load inside wgrad / dgrad
        elif needs_transform:
            arows = L.load().amx_rows_for(npix)
            bias_part = _empty((arows, cos), a) if (has_bias and self.mask is None) else None
            dpre = _empty(a.shape, a)
            kk = (L.ptr(k[0]), L.ptr(k[1]), L.ptr(k[2])) if k is not None else (None, None, None)
            L.call("amx_bn_bwd_apply", L.ptr(dy), L.ptr(a), L.ptr(out.gx), *kk, self.slope, npix, cos,
                   L.ptr(dpre), L.ptr(bias_part), sp)
            k = None
            if self.mask is not None:
                # d conv = mask * lrelu'(.) * (...): the saved activation is already masked, so the sign test of
                # bn_bwd_apply is right wherever the mask is non-zero, and the mask zeroes the rest
                L.call("amx_dropout_bwd", L.ptr(dpre), L.ptr(self.mask), dpre.numel(), sp)
                if out.gx is not None:        # DilatedBlock: the convolution output itself is a summed sub-layer too
                    L.call("amx_add_inplace", L.ptr(dpre), L.ptr(out.gx), dpre.numel(), sp)
                if has_bias:                                  # bias gradient = column sums of the masked dpre
                    nch = 128 if npix >= 1024 else 1
                    tmp = _empty((nch, cos), a)
                    L.call("amx_reduce_rows_chunked", L.ptr(dpre), npix, cos, nch, L.ptr(tmp), sp)
                    db = grad_buffer(self.conv.bias, a)
                    L.call("amx_reduce_rows", L.ptr(tmp), -(-npix // -(-npix // nch)), cos, self.cout, 1.0, L.ptr(db), sp)
                    tape.add_param_grad(self.conv.bias, db)
                    has_bias = False                          # done: neither bias_part nor the wgrad kernel adds it
        else:
            assert self.mask is None, "dropout without an activation is not a layer of the reference's blocks"
            dpre = dy                                       # linear convolution (1x1 of UpsampleBlock)
        if bias_part is not None:
            db = grad_buffer(self.conv.bias, a)
            L.call("amx_reduce_rows", L.ptr(bias_part), bias_part.shape[0], cos, self.cout, 1.0, L.ptr(db), sp)
            tape.add_param_grad(self.conv.bias, db)
        kptr = (k[0], k[1], k[2]) if k is not None else (None, None, None)
        w = self.conv.weight
        dw = grad_buffer(w, a)
        want_bias = has_bias and bias_part is None
        # The weight gradient (MFMA-bound) has no consumer inside backward: side stream.
        w_in = (dpre, aux, kptr)
        d_in = (dpre, aux, kptr) if dpre_mat is None else (dpre_mat, None, (None, None, None))
        # (enqueued BEFORE the layer's data gradient; after it, or on the main stream, measured no different / slower:
        #  profiles/r04_wgrad_ws.md.  Tape.use_side_stream = False serialises everything for the per-kernel timing pass.)
        if self.x_plain is not None and FIRST_WGRAD_MAIN:
            # The net's first layer is the LAST node of backward: nothing follows on the main stream, while the side stream
            # still drains its backlog of weight gradients — this one runs on the (idle) main stream beside them (round 6:
            # the step used to end with conv1_wgrad queued behind c2.0's weight gradient, profiles/r06_step_timeline_final.txt)
            self._wgrad(tape, w_in[0], w_in[1], w_in[2], dw, a, want_bias)
            return
        with tape.side(a, keep=(dpre, dy, a, k, dpre_mat)):
            self._wgrad(tape, w_in[0], w_in[1], w_in[2], dw, a, want_bias)
        if self.x_plain is not None:
            return
        s0 = self.srcs[0]
        s1 = self.srcs[1] if len(self.srcs) > 1 else None
        N, H, W = s0.N, s0.H, s0.W
        C0, C0s = s0.C, s0.Cs
        C1, C1s = (s1.C, s1.Cs) if s1 else (0, 0)
        self._dgrad(tape, d_in[0], d_in[1], d_in[2], s0, s1, N, H, W, C0, C0s, C1, C1s, cos, sp)

    def _bwd_fusable(self, out) -> bool:
        """True when this layer's BatchNorm / LeakyReLU backward can be formed inside the loaders of its two consumers
        (round 4): the data gradient must be a launch of the wave-specialised kernel writing fresh gradients (no
        accumulation into an existing one), nothing else may be folded into dpre (dropout mask, DilatedBlock sum)."""
        if self.x_plain is not None or self.mask is not None or out.gx is not None or self.post_slope != 1.0:
            return False
        if not self.srcs or len(self.srcs) > 2:
            return False
        s0 = self.srcs[0]
        s1 = self.srcs[1] if len(self.srcs) > 1 else None
        if not (s0.needs_grad or (s1 is not None and s1.needs_grad)):
            return False
        for s in (s0, s1):
            if s is not None and (not s.needs_grad or s.grad is not None or s.gx is not None):
                return False
        lib = L.load()
        self._dgrad_split = False
        if lib.amx_conv2d_dgrad_fused_supported(out.Cs, s0.Cs, s1.Cs if s1 else 0, s0.N, s0.H, s0.W, self.taps, self.dil):
            return True
        # Round 6: a layer that read a CONCATENATION of two sources the kernel takes one at a time (U-Net c5.0: 32 -> 32 + 32)
        # runs its data gradient as TWO launches, one per source, each on the weight image of its half of the input
        # channels — the layer's amx_bn_bwd_apply pass goes, as for the single-launch classes
        if (s1 is not None and DGRAD_SPLIT and s0.C == s0.Cs and s1.C == s1.Cs
                and lib.amx_conv2d_dgrad_fused_supported(out.Cs, s0.Cs, 0, s0.N, s0.H, s0.W, self.taps, self.dil)
                and lib.amx_conv2d_dgrad_fused_supported(out.Cs, s1.Cs, 0, s0.N, s0.H, s0.W, self.taps, self.dil)):
            self._dgrad_split = True
            return True
        return False

    def _wgrad(self, tape, dpre, aux, kptr, dw, a, want_bias) -> None:
        w = self.conv.weight
        cos = self.out.Cs
        sp = _sp(a)
        k1, k2, k3 = kptr

        def colsum(part2d, rows, ncols):
            """[rows][ncols] -> [ncols] in (at most) two deterministic stages."""
            if rows > 64:
                nch = 32 if rows < 1024 else 128
                tmp = _empty((nch, ncols), a)
                L.call("amx_reduce_rows_chunked", L.ptr(part2d), rows, ncols, nch, L.ptr(tmp), sp)
                part2d, rows = tmp, -(-rows // -(-rows // nch))
            out1 = _empty((ncols,), a)
            L.call("amx_reduce_rows_chunked", L.ptr(part2d), rows, ncols, 1, L.ptr(out1), sp)
            return out1

        if self.x_plain is not None:
            x = self.x_plain
            N, _, H, W = x.shape
            if aux is not None:
                if self.out.wg1_parts is not None:
                    # the pooling backward already formed the three sums the gradient is linear in (PoolNode.backward)
                    part3, prows = self.out.wg1_parts
                    self.out.wg1_parts = None
                    ncols = 30 * cos
                    if prows > 16:                       # (the combine kernel walks the rows serially: few, fat chunks)
                        nch = 16
                        tmp = _empty((nch, ncols), a)
                        L.call("amx_reduce_rows_chunked", L.ptr(part3), prows, ncols, nch, L.ptr(tmp), sp)
                        part3, prows = tmp, -(-prows // -(-prows // nch))
                    tot = _empty((10 * cos,), a)
                    L.call("amx_conv1_wgrad_combine", L.ptr(part3), prows, cos, L.ptr(k1), L.ptr(k2), L.ptr(k3), L.ptr(tot), sp)
                else:
                    part = _empty((self.rows, 10, cos), a)
                    L.call("amx_conv1_wgrad_fused", L.ptr(x), L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2), L.ptr(k3),
                           self.slope, L.ptr(part), N, H, W, cos, self.dil, self.rows, self.rows_pix, sp)
                    tot = colsum(part, self.rows, 10 * cos)
                L.call("amx_wgrad_reduce", L.ptr(tot), 1, 9, 1, cos, 1, 1, 0, self.cout, L.ptr(dw), sp)
                if want_bias:
                    db = grad_buffer(self.conv.bias, a)
                    db.copy_(tot[9 * cos: 9 * cos + self.cout])
                    tape.add_param_grad(self.conv.bias, db)
            else:
                part = _empty((self.rows, 9, cos), a)
                L.call("amx_conv1_wgrad", L.ptr(x), L.ptr(dpre), L.ptr(part), N, H, W, cos, self.dil,
                       self.rows, self.rows_pix, sp)
                prows = self.rows
                if prows > 64:           # thousands of per-block rows: chunk sums first (see colsum)
                    nch = 32 if prows < 1024 else 128
                    tmp = _empty((nch, 9 * cos), a)
                    L.call("amx_reduce_rows_chunked", L.ptr(part), prows, 9 * cos, nch, L.ptr(tmp), sp)
                    part, prows = tmp, -(-prows // -(-prows // nch))
                L.call("amx_wgrad_reduce", L.ptr(part), prows, 9, 1, cos, 1, 1, 0, self.cout, L.ptr(dw), sp)
            tape.add_param_grad(w, dw)
            return
        s0 = self.srcs[0]
        s1 = self.srcs[1] if len(self.srcs) > 1 else None
        N, H, W = s0.N, s0.H, s0.W
        C0, C0s = s0.C, s0.Cs
        C1, C1s = (s1.C, s1.Cs) if s1 else (0, 0)
        wrows = L.load().amx_conv2d_wgrad_rows(N, H, W, C0s + C1s, self.cout, self.taps, self.dil)
        ci_pad, co_pad = r16(C0s + C1s), r16(self.cout)
        part = _empty((wrows, self.taps, ci_pad, co_pad), a)
        bpart = None
        if want_bias:
            ks = L.load().amx_conv2d_wgrad_ksplit(N, H, W, C0s + C1s, self.cout, self.taps, self.dil)
            bpart = _empty((ks, co_pad), a)
        ps0, ps1 = s0.post_slope, (s1.post_slope if s1 else 1.0)
        if ps0 != 1.0 or ps1 != 1.0:
            L.call("amx_conv2d_wgrad_act", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), ps0, C0s,
                   L.ptr(s1.t if s1 else None), L.ptr(s1.scale if s1 else None),
                   L.ptr(s1.shift if s1 else None), ps1, C1s, L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2),
                   L.ptr(k3), self.slope, cos, L.ptr(part), L.ptr(bpart), N, H, W, self.cout, self.taps,
                   self.dil, sp)
        else:
            L.call("amx_conv2d_wgrad_fused", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), C0s,
                   L.ptr(s1.t if s1 else None), L.ptr(s1.scale if s1 else None),
                   L.ptr(s1.shift if s1 else None), C1s, L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2),
                   L.ptr(k3), self.slope, cos, L.ptr(part), L.ptr(bpart), N, H, W, self.cout, self.taps,
                   self.dil, sp)
        if bpart is not None:
            db = grad_buffer(self.conv.bias, a)
            L.call("amx_reduce_rows", L.ptr(bpart), bpart.shape[0], co_pad, self.cout, 1.0, L.ptr(db), sp)
            tape.add_param_grad(self.conv.bias, db)
        if wrows > 8:                    # two-stage: coalesced (1 KB per row and block) chunk sums first;
            #                              the strided 128 B reads of wgrad_reduce are ~10x slower per byte
            nch = 8 if wrows <= 64 else (32 if wrows < 1024 else 128)
            ncols = self.taps * ci_pad * co_pad
            part2 = _empty((nch, ncols), a)
            L.call("amx_reduce_rows_chunked", L.ptr(part), wrows, ncols, nch, L.ptr(part2), sp)
            part, wrows = part2, -(-wrows // -(-wrows // nch))
        L.call("amx_wgrad_reduce", L.ptr(part), wrows, self.taps, ci_pad, co_pad, C0, C0s, C1, self.cout,
               L.ptr(dw), sp)
        tape.add_param_grad(w, dw)

    def _dgrad(self, tape, dpre, aux, kptr, s0, s1, N, H, W, C0, C0s, C1, C1s, cos, sp) -> None:
        # ---- data gradient(s): forward conv of dpre with the flipped / transposed weight image
        w = self.conv.weight
        k1, k2, k3 = kptr
        need0 = s0.needs_grad
        need1 = bool(s1 and s1.needs_grad)
        if not (need0 or need1):
            return
        split = aux is not None and s1 is not None and getattr(self, "_dgrad_split", False)
        wpk = None if split else pack_weights(w, C0, C0s, C1, C1s, self.taps, 1)
        bsum_src = s1 is None and s0.wants_bsum_from_dgrad(self, tape.training)      # (asked before s0.grad is allocated)
        tgt = []
        for s in (s0, s1):
            if s is None:
                tgt.append((None, None))
                continue
            if s.grad is None:
                s.grad = _empty(s.t.shape, s.t)
                tgt.append((s.grad, s.gx))               # first writer; DilatedBlock adds its gx here
            else:
                tgt.append((s.grad, s.grad))             # accumulate in place
        add0 = tgt[0][1]
        if s1 is not None and tgt[1][1] is not None:
            # the kernel accumulates only into its first output: pre-add the second one
            scratch = _empty(s1.t.shape, s1.t)
            y1 = scratch
        else:
            scratch, y1 = None, tgt[1][0]
        if aux is not None:
            # dpre is formed by the loader of the wave-specialised kernel from (dy, a) (_bwd_fusable guarantees support)
            assert add0 is None and scratch is None
            if split:
                for s, off in ((s0, 0), (s1, C0)):
                    wh = pack_weights_range(w, off, s.C, s.Cs, self.taps, 1)
                    L.call("amx_conv2d_dgrad_fused", L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2), L.ptr(k3), self.slope,
                           cos, L.ptr(wh), L.ptr(s.grad), s.Cs, None, 0, N, H, W, self.taps, self.dil, sp)
                return
            if s1 is None and bsum_src:
                # the gradient written here is the whole dy of the source layer: its BatchNorm-backward sums come out of
                # this kernel's epilogue, amx_bn_bwd_reduce is not launched for it (ConvNode.backward reads s0.bstats)
                rows = L.load().amx_conv2d_dgrad_bsum_rows(cos, C0s, N, H, W, self.taps, self.dil)
                if rows > 0:
                    part = _empty((rows, 2, C0s), s0.t)
                    L.call("amx_conv2d_dgrad_fused_bsum", L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2), L.ptr(k3),
                           self.slope, cos, L.ptr(wpk), L.ptr(tgt[0][0]), C0s, N, H, W, self.taps, self.dil, L.ptr(s0.t),
                           L.ptr(part), sp)
                    s0.bstats = (part, rows, C0s, 0)
                    return
            L.call("amx_conv2d_dgrad_fused", L.ptr(dpre), L.ptr(aux), L.ptr(k1), L.ptr(k2), L.ptr(k3), self.slope, cos,
                   L.ptr(wpk), L.ptr(tgt[0][0]), C0s, L.ptr(y1), C1s, N, H, W, self.taps, self.dil, sp)
            return
        L.call("amx_conv2d_dgrad", L.ptr(dpre), cos, L.ptr(wpk), L.ptr(add0), L.ptr(tgt[0][0]), C0s, L.ptr(y1), C1s,
               N, H, W, self.taps, self.dil, sp)
        if scratch is not None:
            L.call("amx_add_inplace", L.ptr(tgt[1][0]), L.ptr(scratch), scratch.numel(), sp)


# UpsampleBlock forward in one pass (csrc/upconv.hip): AMX_FUSE_UPCONV=0 keeps the two launches (host-side switch, read once)
FUSE_UPCONV = _os.environ.get("AMX_FUSE_UPCONV", "1") != "0"
UPCONV_MIN_PIXELS = 1 << 19      # low-res pixels of the launch from which the one-pass kernel wins (profiles/r05_logs/r05_upconv_ab4.log:
#                                  x16 512^2 949 -> 731 us, x32 256^2 253 -> 170, x32 128^2 122 -> 108; x32 64^2 69 vs 73: two kernels)


def upconv_fusable(src: "Act", conv) -> bool:
    """True when UpsampleBlock's 1x1 convolution + x2 interpolation of `src` runs as ONE launch (amx_upconv1x1_fwd):
    a shape the kernel takes (its results are bit-identical to the two-kernel path there) and large enough to pay."""
    if not FUSE_UPCONV or src.post_slope != 1.0 or src.npix < UPCONV_MIN_PIXELS:
        return False
    w = conv.weight
    if w.dtype != torch.float32 or tuple(w.shape[2:]) != (1, 1) or w.shape[1] != src.C:
        return False
    return bool(L.load().amx_upconv1x1_supported(src.C, src.Cs, w.shape[0], r4(w.shape[0])))


class UpConvNode(ConvNode):
    """UpsampleBlock (atomai/nets/blocks.py:86-132) as one node: forward = amx_upconv1x1_fwd (1x1 convolution at low
    resolution + x2 interpolation, the low-resolution tensor never written); backward = amx_upsample2x_bwd, then the
    ordinary 1x1 weight / data gradients of ConvNode on the low-resolution gradient — exactly the launches of the
    UpsampleNode + ConvNode pair it replaces."""

    def __init__(self, tape, src: Act, conv, mode: str):
        self.up_mode = {"bilinear": 0, "nearest": 1}[mode]
        super().__init__(tape, [src], conv, None, 1.0)

    def _forward(self, tape) -> Act:
        s0 = self.srcs[0]
        cos = r4(self.cout)
        y = _empty((s0.N, 2 * s0.H, 2 * s0.W, cos), s0.t)
        w, b = self.conv.weight, self.conv.bias
        L.call("amx_upconv1x1_fwd", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), L.ptr(w.detach().contiguous()),
               L.ptr(b.detach() if b is not None else None), L.ptr(y), s0.N, s0.H, s0.W, s0.C, s0.Cs, self.cout, cos,
               self.up_mode, _sp(y))
        return Act(y, self.cout, needs_grad=tape.need_grad)

    def backward(self, tape) -> None:
        hi = self.out
        g = hi.grad if hi.grad is not None else hi.gx
        if g is None:
            return
        s0 = self.srcs[0]
        # the stand-in below carries a GRADIENT where ConvNode.backward expects an activation: safe only because a linear
        # convolution (no BatchNorm, no LeakyReLU, no dropout) never reads its own output in backward
        assert self.bn is None and self.slope == 1.0 and self.post_slope == 1.0 and self.mask is None
        dv = _empty((s0.N, s0.H, s0.W, hi.Cs), g)
        L.call("amx_upsample2x_bwd", L.ptr(g), L.ptr(dv), s0.N, s0.H, s0.W, hi.Cs, self.up_mode, _sp(g))
        lo = Act(dv, self.cout, needs_grad=True)             # the low-resolution convolution output's stand-in: only its
        lo.grad = dv                                          # gradient exists (the tensor itself never did)
        self.out = lo
        try:
            super().backward(tape)
        finally:
            self.out = hi


class ResOutNode(_Node):
    """Tail of a ResBlock (atomai/nets/blocks.py:210-213): out = LeakyReLU(bn2(t) + r), materialised.
    ``t`` carries bn2 as its pending affine; ``r`` is the block's c0 output (plain tensor)."""

    def __init__(self, tape, t: Act, r: Act, slope: float):
        assert t.post_slope == 1.0 and r.post_slope == 1.0 and r.scale is None
        assert t.t.shape == r.t.shape
        self.t_act, self.r_act, self.slope = t, r, float(slope)
        t.consumed_by(self)
        r.consumed_by(self)
        y = _empty(t.t.shape, t.t)
        L.call("amx_res_out_fwd", L.ptr(t.t), L.ptr(t.scale), L.ptr(t.shift), L.ptr(r.t), self.slope, t.npix, t.Cs,
               L.ptr(y), _sp(y))
        self.out = Act(y, t.C, needs_grad=tape.need_grad)

    def backward(self, tape) -> None:
        g = self.out.grad
        if g is None:
            return
        t, r = self.t_act, self.r_act
        ds = _empty(g.shape, g)
        # the convolution branch gets its own copy when its producer forwards the gradient tensor unchanged to a
        # weight-gradient kernel on the side stream (no BatchNorm in between) while the residual branch keeps
        # accumulating into it.  With a BatchNorm in between the same hazard exists whenever the producer's backward is
        # formed inside its consumers' loaders (ConvNode._bwd_fusable): the side stream's weight-gradient kernel then
        # reads THIS tensor as `dy` while the residual branch's data gradient adds into it on the main stream.
        alias_unsafe = (not t.producer) or bwd_fuse_enabled()
        ds2 = _empty(g.shape, g) if (alias_unsafe and r.needs_grad) else None
        L.call("amx_lrelu_bwd", L.ptr(g), L.ptr(self.out.t), None, None, self.slope, t.npix, t.Cs, L.ptr(ds),
               L.ptr(ds2), _sp(g))
        tape.accumulate(t, ds)
        if r.needs_grad:
            tape.accumulate(r, ds2 if ds2 is not None else ds)


class PoolNode(_Node):
    def __init__(self, tape, src: Act):
        self.src = src
        src.consumed_by(self)
        fused = src.pooled
        if fused is not None:                        # eval mode: produced by the first-layer kernel itself
            self.out = fused
            return
        y = _empty((src.N, src.H // 2, src.W // 2, src.Cs), src.t)
        L.call("amx_pool2x2_fwd", L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(y), src.N,
               src.H, src.W, src.Cs, _sp(y))
        self.out = Act(y, src.C, needs_grad=src.needs_grad)

    def backward(self, tape) -> None:
        s, g = self.src, self.out.grad
        if g is None or not s.needs_grad:
            return
        skip = s.grad
        if (s.wg1 is not None and s.gx is None and s.wants_bstats(self, tape.training)
                and L.load().amx_pool2x2_bwd_wgrad1_supported(s.H, s.W, s.Cs, 1)):
            # first layer of the net: its weight-gradient sums come out of this pass, dy itself is not written (see the kernel)
            x_plain, slope = s.wg1
            rows = L.load().amx_pool2x2_bwd_rows(s.N, s.H, s.W, s.Cs)
            bstats = _empty((rows, 2, s.Cs), g)
            part3 = _empty((rows, 3, 10, s.Cs), g)
            L.call("amx_pool2x2_bwd_wgrad1", L.ptr(g), L.ptr(s.t), L.ptr(s.scale), L.ptr(s.shift), L.ptr(skip),
                   L.ptr(x_plain), slope, L.ptr(bstats), L.ptr(part3), s.N, s.H, s.W, s.Cs, _sp(g))
            s.bstats = (bstats, rows, s.Cs, 0)
            s.wg1_parts = (part3, rows)
            if s.grad is None:
                s.grad = g                                # (any tensor: the producer's backward only asks "is there a gradient")
            return
        if s.grad is None:
            s.grad = _empty(s.t.shape, s.t)
            skip = s.gx
        bstats = None
        if s.wants_bstats(self, tape.training) and 256 % (s.Cs // 4) == 0:
            rows = L.load().amx_pool2x2_bwd_rows(s.N, s.H, s.W, s.Cs)
            bstats = _empty((rows, 2, s.Cs), g)
            s.bstats = (bstats, rows, s.Cs, 0)
        L.call("amx_pool2x2_bwd", L.ptr(g), L.ptr(s.t), L.ptr(s.scale), L.ptr(s.shift), L.ptr(skip),
               L.ptr(s.grad), L.ptr(bstats), s.N, s.H, s.W, s.Cs, _sp(g))


class UpsampleNode(_Node):
    def __init__(self, tape, src: Act, mode: str):
        assert src.scale is None, "upsample expects a materialised (affine-free) activation"
        self.src, self.mode = src, {"bilinear": 0, "nearest": 1}[mode]
        src.consumed_by(self)
        y = _empty((src.N, 2 * src.H, 2 * src.W, src.Cs), src.t)
        L.call("amx_upsample2x_fwd", L.ptr(src.t), L.ptr(y), src.N, src.H, src.W, src.Cs, self.mode, _sp(y))
        self.out = Act(y, src.C, needs_grad=src.needs_grad)

    def backward(self, tape) -> None:
        s, g = self.src, self.out.grad
        if g is None or not s.needs_grad:
            return
        dv = _empty(s.t.shape, s.t)
        L.call("amx_upsample2x_bwd", L.ptr(g), L.ptr(dv), s.N, s.H, s.W, s.Cs, self.mode, _sp(g))
        tape.accumulate(s, dv)


class ResizeCatNode(_Node):
    """torch.cat([F.interpolate(s, size=(H, W), mode) for s in srcs], 1) for small-channel score maps
    (ResHedNet.forward, atomai/nets/fcnn.py:283-295); each source's pending BatchNorm affine is applied on load."""

    def __init__(self, tape, srcs: Sequence[Act], H: int, W: int, mode: str):
        self.srcs, self.mode = list(srcs), {"bilinear": 0, "nearest": 1}[mode]
        a0 = self.srcs[0]
        C = sum(s.C for s in self.srcs)
        y = torch.zeros((a0.N, H, W, r4(C)), dtype=torch.float32, device=a0.t.device)
        off = 0
        for s_ in self.srcs:
            assert s_.post_slope == 1.0
            s_.consumed_by(self)
            L.call("amx_resize_cat_fwd", L.ptr(s_.t), L.ptr(s_.scale), L.ptr(s_.shift), s_.N, s_.H, s_.W, s_.Cs, s_.C,
                   L.ptr(y), H, W, r4(C), off, self.mode, _sp(y))
            off += s_.C
        self.out = Act(y, C, needs_grad=any(s_.needs_grad for s_ in self.srcs))

    def backward(self, tape) -> None:
        g = self.out.grad
        if g is None:
            return
        off = 0
        for s_ in self.srcs:
            if s_.needs_grad:
                ds = _empty(s_.t.shape, s_.t)
                L.call("amx_resize_cat_bwd", L.ptr(g), self.out.N, self.out.H, self.out.W, self.out.Cs, off, s_.C,
                       L.ptr(ds), s_.H, s_.W, s_.Cs, self.mode, _sp(g))
                tape.accumulate(s_, ds)
            off += s_.C


class DilatedSumNode(_Node):
    """out = sum_i (pre_i + a_i + bn_i) over the layers of a DilatedBlock (blocks.py:321-329).  With a Dropout layer in
    the block (blocks.py:311-312) its output is one more summed sub-layer: in eval mode it equals pre_i (wpre = 2); in
    training pre_i / a_i / bn_i are those of the DROPPED tensor and the un-dropped convolution output is added by a
    second pass over ``act.unmasked``."""

    def __init__(self, tape, acts: Sequence[Act], slope: float, wpre: float = 1.0):
        self.acts, self.slope = list(acts), slope
        for act in self.acts:
            act.consumed_by(self)
        a0 = acts[0]
        y = _empty(a0.t.shape, a0.t)
        PP = ctypes.c_void_p * 4
        for i in range(0, len(acts), 4):
            grp = acts[i:i + 4]
            pa = PP(*[x.t.data_ptr() for x in grp] + [0] * (4 - len(grp)))
            ps = PP(*[(x.scale.data_ptr() if x.scale is not None else 0) for x in grp] + [0] * (4 - len(grp)))
            ph = PP(*[(x.shift.data_ptr() if x.shift is not None else 0) for x in grp] + [0] * (4 - len(grp)))
            L.call("amx_dilated_sum_ex", pa, ps, ph, len(grp), slope, 1 if i else 0, float(wpre), 1.0, L.ptr(y),
                   a0.npix, a0.Cs, _sp(y))
        um = [x.unmasked for x in acts if x.unmasked is not None]
        for i in range(0, len(um), 4):
            grp = um[i:i + 4]
            pa = PP(*[t.data_ptr() for t in grp] + [0] * (4 - len(grp)))
            zero = PP(0, 0, 0, 0)
            L.call("amx_dilated_sum_ex", pa, zero, zero, len(grp), slope, 1, 1.0, 0.0, L.ptr(y), a0.npix, a0.Cs, _sp(y))
        for x in acts:
            x.unmasked = None                            # read once; nothing in backward needs it
        self.out = Act(y, a0.C, needs_grad=a0.needs_grad)

    def backward(self, tape) -> None:
        g = self.out.grad
        if g is None:
            return
        for a in self.acts:
            assert a.grad is None and a.gx is None
            a.gx = g


class InputNode(_Node):
    """NCHW tensor at the module boundary -> NHWC activation."""

    def __init__(self, tape, x: torch.Tensor):
        N, C, H, W = x.shape
        self.x_shape = x.shape
        cs = r4(C)
        t = _empty((N, H, W, cs), x)
        L.call("amx_nchw_to_nhwc", L.ptr(x.detach().contiguous()), L.ptr(t), N, C, cs, H, W, _sp(x))
        self.out = Act(t, C, needs_grad=bool(x.requires_grad and tape.need_grad))
        self.grad_nchw = None

    def backward(self, tape) -> None:
        a = self.out
        g = a.grad if a.grad is not None else a.gx
        if not a.needs_grad or g is None:
            return
        N, C, H, W = self.x_shape
        out = _empty(self.x_shape, a.t)
        L.call("amx_nhwc_to_nchw", L.ptr(g), L.ptr(out), N, C, a.Cs, H, W, _sp(out))
        self.grad_nchw = out


class OutputNode(_Node):
    """NHWC activation (pending affine materialised) -> NCHW tensor at the module boundary."""

    def __init__(self, tape, src: Act):
        self.src = src
        src.consumed_by(self)
        t = src.t
        if src.scale is not None:
            t = _empty(src.t.shape, src.t)
            L.call("amx_affine_nhwc", L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(t), src.npix,
                   src.Cs, _sp(t))
        self.value = _empty((src.N, src.C, src.H, src.W), t)
        L.call("amx_nhwc_to_nchw", L.ptr(t), L.ptr(self.value), src.N, src.C, src.Cs, src.H, src.W, _sp(t))
        self.grad_out: Optional[torch.Tensor] = None

    def backward(self, tape) -> None:
        s = self.src
        if self.grad_out is None or not s.needs_grad:
            return
        g = _empty(s.t.shape, s.t)
        L.call("amx_nchw_to_nhwc", L.ptr(self.grad_out.contiguous()), L.ptr(g), s.N, s.C, s.Cs, s.H, s.W,
               _sp(g))
        tape.accumulate(s, g)


# experiment switch for the eval-mode epilogue fusions (classification head, DilatedBlock sum); 0 = separate kernels
FUSE_HEAD = _os.environ.get("AMX_FUSE_HEAD", "1") != "0"
# Eval mode: the 2x2 max-pool that follows a one-layer first block (Unet / dilnet c1) is produced by the first-layer kernel
# (amx_conv1_fwd_pool); AMX_FUSE_POOL=0 keeps the separate amx_pool2x2_fwd launch (A/B switch, tools/gpu_step_ab.py).
FUSE_POOL = _os.environ.get("AMX_FUSE_POOL", "1") != "0"


def head_fusable(tape, srcs: Sequence[Act], conv, px) -> bool:
    """Can `px` (the net's final 1x1 convolution) be evaluated in the epilogue of `conv` (its last 3x3 layer)?  Eval
    mode only (in training the layer's BatchNorm statistics do not exist before the layer has been computed)."""
    if not FUSE_HEAD or tape.training or tape.need_grad or px.weight.shape[0] > 3:
        return False
    if tuple(conv.kernel_size) != (3, 3) or tuple(conv.dilation) != (1, 1) or tuple(px.kernel_size) != (1, 1):
        return False
    if any(s.post_slope != 1.0 for s in srcs):
        return False
    cin_s = sum(s.Cs for s in srcs)
    return bool(L.load().amx_conv2d_head_supported(cin_s, conv.weight.shape[0], 9, 1, srcs[0].H))


class HeadNode(_Node):
    """Last 3x3 layer of a net in eval mode with the classification head fused into its epilogue
    (amx_conv2d_fwd_head): value = logits NCHW (mode 0) or probabilities NHWC (mode 1); the layer's activation never
    reaches HBM."""

    def __init__(self, tape, srcs: Sequence[Act], conv, bn, slope: float, px, mode: int):
        s0 = srcs[0]
        s1 = srcs[1] if len(srcs) > 1 else None
        N, H, W = s0.N, s0.H, s0.W
        C0, C0s = s0.C, s0.Cs
        C1, C1s = (s1.C, s1.Cs) if s1 else (0, 0)
        w, b = conv.weight, conv.bias
        cout, K = w.shape[0], px.weight.shape[0]
        cos, cop = r4(cout), r16(cout)
        assert w.shape[1] == C0 + C1 and px.weight.shape[1] == cout
        wpk = pack_weights(w, C0, C0s, C1, C1s, 9, 0)

        def fold():
            Wp = px.weight.detach().reshape(K, cout)
            hw = torch.zeros((K, cop), dtype=torch.float32, device=s0.t.device)
            if bn is not None:                               # fold the layer's own eval-mode affine into the head
                scale, shift = bn_eval_affine(bn, cout, cos, s0.t)
                hw[:, :cout] = Wp * scale[:cout]
                hb = (px.bias.detach() + (Wp * shift[:cout]).sum(1)).contiguous()
            else:
                hw[:, :cout] = Wp
                hb = px.bias.detach().contiguous()
            return hw, hb
        deps = (px.weight, px.bias) + ((bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else ())
        hw, hb = cached_eval(px, deps, ("head", id(bn), cop, s0.t.device.index), fold)
        shape = (N, K, H, W) if mode == 0 else (N, H, W, K)
        self.value = _empty(shape, s0.t)
        L.call("amx_conv2d_fwd_head", L.ptr(s0.t), L.ptr(s0.scale), L.ptr(s0.shift), C0s,
               L.ptr(s1.t if s1 else None), L.ptr(s1.scale if s1 else None), L.ptr(s1.shift if s1 else None), C1s,
               L.ptr(wpk), L.ptr(b.detach() if b is not None else None), L.ptr(hw), L.ptr(hb), L.ptr(self.value),
               K, mode, N, H, W, cout, float(slope), _sp(s0.t))
        self.grad_out = None

    def backward(self, tape) -> None:
        raise AssertionError("the fused head exists in eval mode only")


def dsum_fusable(tape, srcs: Sequence[Act], conv, acts: Sequence[Act]) -> bool:
    """Can the sum of a DilatedBlock be evaluated in the epilogue of `conv`, the block's last layer?  Eval mode only (in
    training the last layer's activation is needed by its own backward and must be stored anyway)."""
    if not FUSE_HEAD or tape.training or tape.need_grad or not (1 <= len(acts) <= 3) or len(srcs) != 1:
        return False
    if tuple(conv.kernel_size) != (3, 3) or srcs[0].post_slope != 1.0:
        return False
    return bool(L.load().amx_conv2d_dsum_supported(srcs[0].Cs, conv.weight.shape[0], 9, int(conv.dilation[0]), srcs[0].H))


class DsumConvNode(_Node):
    """Last layer of a DilatedBlock in eval mode with the block's output (the sum of every sub-layer output,
    blocks.py:321-329) fused into its epilogue (amx_conv2d_fwd_dsum); the layer's own activation is not stored."""

    def __init__(self, tape, src: Act, conv, bn, slope: float, acts: Sequence[Act]):
        N, H, W = src.N, src.H, src.W
        w, b = conv.weight, conv.bias
        cout = w.shape[0]
        cos = r4(cout)
        wpk = pack_weights(w, src.C, src.Cs, 0, 0, 9, 0)
        zero = torch.zeros((cos,), dtype=torch.float32, device=src.t.device)
        if bn is not None:
            scale, shift = bn_eval_affine(bn, cout, cos, src.t)
        else:
            scale = shift = zero
        scs = [(a.scale if a.scale is not None else zero) for a in acts] + [scale]
        shs = [(a.shift if a.shift is not None else zero) for a in acts] + [shift]
        n = len(acts)
        PA, PS = ctypes.c_void_p * n, ctypes.c_void_p * (n + 1)
        y = _empty((N, H, W, cos), src.t)
        self.keep = (scs, shs, zero, [a.t for a in acts])     # raw data_ptr() arrays below: keep every tensor alive
        L.call("amx_conv2d_fwd_dsum", L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), src.Cs, L.ptr(wpk),
               L.ptr(b.detach() if b is not None else None), PA(*[a.t.data_ptr() for a in acts]),
               PS(*[t.data_ptr() for t in scs]), PS(*[t.data_ptr() for t in shs]), n, L.ptr(y), N, H, W, cout,
               int(conv.dilation[0]), float(slope), _sp(src.t))
        self.out = Act(y, cout)

    def backward(self, tape) -> None:
        raise AssertionError("the fused DilatedBlock sum exists in eval mode only")


class PxNode(_Node):
    """Final 1x1 conv to nb_classes; logits NCHW (mode 0) or probabilities NHWC (mode 1)."""

    def __init__(self, tape, src: Act, conv, mode: int = 0):
        self.src, self.conv = src, conv
        src.consumed_by(self)
        K = conv.weight.shape[0]
        assert conv.weight.shape[1] == src.C and conv.weight.shape[2:] == (1, 1)
        self.K = K
        shape = (src.N, K, src.H, src.W) if mode == 0 else (src.N, src.H, src.W, K)
        self.value = _empty(shape, src.t)
        L.call("amx_px_fwd", L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift),
               L.ptr(conv.weight.detach()), L.ptr(conv.bias.detach()), L.ptr(self.value), src.N, src.H,
               src.W, src.C, src.Cs, K, mode, _sp(src.t))
        self.grad_out: Optional[torch.Tensor] = None

    def backward(self, tape) -> None:
        s, dl = self.src, self.grad_out
        if dl is None:
            return
        dl = dl.contiguous()
        rows = L.load().amx_rows_for(s.npix)
        rows_pix = L.load().amx_rows_pix(s.npix)
        dxn = _empty(s.t.shape, s.t)
        part = _empty((rows, self.K, s.Cs), s.t)
        partb = _empty((rows, self.K), s.t)
        sp = _sp(dl)
        bstats = None
        if s.wants_bstats(self, tape.training) and s.grad is None:
            bstats = _empty((rows, 2, s.Cs), s.t)
            s.bstats = (bstats, rows, s.Cs, 0)
        L.call("amx_px_bwd", L.ptr(dl), L.ptr(s.t), L.ptr(s.scale), L.ptr(s.shift),
               L.ptr(self.conv.weight.detach()), L.ptr(dxn), L.ptr(part), L.ptr(partb), L.ptr(bstats), s.N, s.H,
               s.W, s.C, s.Cs, self.K, rows, rows_pix, sp)
        dw = _empty((self.K * s.Cs,), s.t)
        L.call("amx_reduce_rows", L.ptr(part), rows, self.K * s.Cs, self.K * s.Cs, 1.0, L.ptr(dw), sp)
        db = grad_buffer(self.conv.bias, s.t)
        L.call("amx_reduce_rows", L.ptr(partb), rows, self.K, self.K, 1.0, L.ptr(db), sp)
        tape.add_param_grad(self.conv.weight, dw.view(self.K, s.Cs)[:, : s.C].reshape(self.K, s.C, 1, 1))
        tape.add_param_grad(self.conv.bias, db)
        if s.needs_grad:
            tape.accumulate(s, dxn)


class PxLossNode(_Node):
    """The training step's head in ONE pass (amx_px_ce_train): px -> CrossEntropyLoss (BCEWithLogitsLoss for one class) ->
    their backward, for the trainers'
    fused step (nets/fcnn.py: forward_loss).  value = the scalar mean loss; what px_bwd would have produced for an upstream
    gradient of 1 is kept for backward, which only scales it if the upstream gradient is not 1."""

    def __init__(self, tape, src: Act, conv, target: torch.Tensor):
        self.src, self.conv = src, conv
        src.consumed_by(self)
        K = conv.weight.shape[0]
        assert conv.weight.shape[1] == src.C and conv.weight.shape[2:] == (1, 1)
        assert target.numel() == src.npix and target.dtype == (torch.float32 if K == 1 else torch.int64)
        self.K = K
        s = src
        rows = L.load().amx_rows_for(s.npix)
        rows_pix = L.load().amx_rows_pix(s.npix)
        need = tape.need_grad
        self.dxn = _empty(s.t.shape, s.t)
        self.part = _empty((rows, K, s.Cs), s.t)
        self.partb = _empty((rows, K), s.t)
        self.bstats = None
        if need and s.wants_bstats(self, tape.training):
            self.bstats = _empty((rows, 2, s.Cs), s.t)
        lpart = _empty((rows,), s.t)
        sp = _sp(s.t)
        L.call("amx_px_ce_train", L.ptr(s.t), L.ptr(s.scale), L.ptr(s.shift), L.ptr(conv.weight.detach()),
               L.ptr(conv.bias.detach()), L.ptr(target.contiguous() if K > 1 else None),
               L.ptr(target.contiguous() if K == 1 else None), L.ptr(self.dxn), L.ptr(self.part), L.ptr(self.partb),
               L.ptr(self.bstats), L.ptr(lpart), s.N, s.H, s.W, s.C, s.Cs, K, rows, rows_pix, sp)
        self.value = torch.empty((), dtype=torch.float32, device=s.t.device)
        L.call("amx_reduce_rows", L.ptr(lpart), rows, 1, 1, 1.0 / s.npix, L.ptr(self.value), sp)
        self.rows = rows
        self.grad_out: Optional[torch.Tensor] = None

    def backward(self, tape) -> None:
        s, g = self.src, self.grad_out
        if g is None:
            return
        dxn, part, partb, bstats = self.dxn, self.part, self.partb, self.bstats
        self.dxn = self.part = self.partb = self.bstats = None
        sp = _sp(dxn)
        if g.numel() == 1 and g.dtype == torch.float32 and g.device == dxn.device:
            L.call("amx_scale_unless_one_multi", L.ptr(dxn), dxn.numel(), L.ptr(part), part.numel(), L.ptr(partb),
                   partb.numel(), L.ptr(bstats), bstats.numel() if bstats is not None else 0, L.ptr(g.contiguous()), sp)
        else:
            gs = g.to(dxn.dtype).reshape(())
            dxn, part, partb = dxn * gs, part * gs, partb * gs
            bstats = bstats * gs if bstats is not None else None
        if bstats is not None and s.grad is None:
            s.bstats = (bstats, self.rows, s.Cs, 0)
        dw = _empty((self.K * s.Cs,), s.t)
        L.call("amx_reduce_rows", L.ptr(part), self.rows, self.K * s.Cs, self.K * s.Cs, 1.0, L.ptr(dw), sp)
        db = grad_buffer(self.conv.bias, s.t)
        L.call("amx_reduce_rows", L.ptr(partb), self.rows, self.K, self.K, 1.0, L.ptr(db), sp)
        tape.add_param_grad(self.conv.weight, dw.view(self.K, s.Cs)[:, : s.C].reshape(self.K, s.C, 1, 1))
        tape.add_param_grad(self.conv.bias, db)
        if s.needs_grad:
            tape.accumulate(s, dxn)


def px_loss_fusable(src: Act, conv, target) -> bool:
    K = conv.weight.shape[0]
    return (FUSE_PX_LOSS and isinstance(target, torch.Tensor) and target.numel() == src.npix
            and target.dtype == (torch.float32 if K == 1 else torch.int64)
            and bool(L.load().amx_px_ce_train_supported(src.Cs, K)))


FUSE_PX_LOSS = _os.environ.get("AMX_FUSE_PX_LOSS", "1") != "0"


# ====================================================================================== tape
_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def aux_stream(dev, which: int = 0) -> "torch.cuda.Stream":
    """Persistent auxiliary stream number `which` of a device (0: weight gradients, 1 / 2: the predictor's
    upload / download copies).  Persistent so that its hardware queue never changes between calls."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get((idx, which))
    if st is None:
        st = _SIDE_STREAMS[(idx, which)] = torch.cuda.Stream(dev)
    return st


def _side_stream(dev) -> "torch.cuda.Stream":
    """ONE persistent side stream per device.  The HIP runtime multiplexes streams onto 4 hardware queues round-robin;
    with a fresh stream per step every 4th one landed on the main stream's queue and that step lost the
    weight-gradient overlap (+2.6 ms, visible as a period-4 pattern in the per-step times)."""
    return aux_stream(dev, 0)


class _SideCtx:
    """Runs the enclosed launches on the tape's side stream, ordered after everything issued so far on the main
    stream; tensors in `keep` stay referenced until the streams are joined at the end of backward."""

    def __init__(self, tape, like, keep):
        self.tape, self.like, self.keep = tape, like, keep
        self.ctx = None

    def __enter__(self):
        t = self.tape
        if not (self.like.is_cuda and t.use_side_stream):
            return self
        dev = self.like.device
        if t.side_stream is None:
            t.side_stream = _side_stream(dev)
        main = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        t.side_stream.wait_event(ev)
        t.keepalive.extend(self.keep)
        self.ctx = torch.cuda.stream(t.side_stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class Tape:
    use_side_stream = True

    def __init__(self, training: bool, need_grad: bool):
        self.training = training
        self.need_grad = need_grad
        self.nodes: List[_Node] = []
        self.param_grads: Dict[int, tuple] = {}
        self.side_stream = None
        self.keepalive: list = []
        self.bn_counters: list = []
        # predictor: (min, ptp) of the stack; the first-layer kernel reads its input as (x - min) / ptp
        self.input_norm = None
        self.input_norm_used = False

    def side(self, like: torch.Tensor, keep=()):
        return _SideCtx(self, like, keep)

    def end_forward(self) -> None:
        """BatchNorm's `num_batches_tracked += 1` of every layer of this forward pass as one multi-tensor launch
        (13 separate 5-microsecond kernels sat between the convolutions of the forward chain otherwise)."""
        if self.bn_counters:
            torch._foreach_add_(self.bn_counters, 1)
            self.bn_counters = []

    # ---- graph construction (each call launches the forward kernels immediately)
    def _push(self, node):
        if self.need_grad:
            self.nodes.append(node)
        return node

    def input(self, x: torch.Tensor) -> InputNode:
        return self._push(InputNode(self, x))

    def conv(self, srcs, conv, bn=None, slope: float = 1.0, post_slope: float = 1.0, drop_p: float = 0.0,
             keep_unmasked: bool = False) -> Act:
        return self._push(ConvNode(self, srcs, conv, bn, slope, post_slope=post_slope, drop_p=drop_p,
                                   keep_unmasked=keep_unmasked)).out

    def res_out(self, t: Act, r: Act, slope: float) -> Act:
        return self._push(ResOutNode(self, t, r, slope)).out

    def conv_first(self, x_plain: torch.Tensor, conv, bn=None, slope: float = 1.0, drop_p: float = 0.0,
                   pool_next: bool = False) -> Act:
        return self._push(ConvNode(self, [], conv, bn, slope, x_plain=x_plain.detach().contiguous(),
                                   drop_p=drop_p, pool_next=pool_next)).out

    def pool(self, src: Act) -> Act:
        return self._push(PoolNode(self, src)).out

    def upsample(self, src: Act, mode: str) -> Act:
        return self._push(UpsampleNode(self, src, mode)).out

    def upconv(self, src: Act, conv, mode: str) -> Act:
        return self._push(UpConvNode(self, src, conv, mode)).out

    def resize_cat(self, srcs, H: int, W: int, mode: str) -> Act:
        return self._push(ResizeCatNode(self, srcs, H, W, mode)).out

    def dilated_sum(self, acts, slope, wpre: float = 1.0) -> Act:
        return self._push(DilatedSumNode(self, acts, slope, wpre)).out

    def output(self, src: Act) -> OutputNode:
        return self._push(OutputNode(self, src))

    def px(self, src: Act, conv, mode: int = 0, loss_target=None):
        """The net's final 1x1 convolution; with `loss_target` (the trainers' fused step, training mode) the mean
        cross-entropy loss against it instead of the logits — one pass, see PxLossNode."""
        if loss_target is not None:
            return self._push(PxLossNode(self, src, conv, loss_target))
        return self._push(PxNode(self, src, conv, mode))

    def conv_dsum(self, src: Act, conv, bn, slope: float, acts) -> Act:
        return DsumConvNode(self, src, conv, bn, slope, acts).out

    def conv_head(self, srcs, conv, bn, slope: float, px, mode: int = 0) -> HeadNode:
        return HeadNode(self, srcs, conv, bn, slope, px, mode)

    # ---- backward helpers
    def accumulate(self, act: Act, g: torch.Tensor) -> None:
        if act.grad is None:
            if act.gx is not None:
                L.call("amx_add_inplace", L.ptr(g), L.ptr(act.gx), g.numel(), _sp(g))
            act.grad = g
        else:
            L.call("amx_add_inplace", L.ptr(act.grad), L.ptr(g), g.numel(), _sp(g))

    def add_param_grad(self, p: torch.Tensor, g: torch.Tensor) -> None:
        key = id(p)
        if key in self.param_grads:
            prev = self.param_grads[key][1]
            gg = g.contiguous()
            n = gg.numel()
            if n % 4 == 0:
                L.call("amx_add_inplace", L.ptr(prev), L.ptr(gg), n, _sp(gg))
            else:
                prev.add_(gg)
        else:
            self.param_grads[key] = (p, g.contiguous())

    def backward(self) -> None:
        for node in reversed(self.nodes):
            node.backward(self)
        self.nodes = []
        if self.side_stream is not None:          # join: gradients written on the side stream are now visible
            ev = torch.cuda.Event()
            ev.record(self.side_stream)
            torch.cuda.current_stream(self.side_stream.device).wait_event(ev)
            # No tensor.record_stream(): the tensors the side stream read were kept referenced (keepalive) until
            # this join, and everything the main stream does from here on is ordered after it — so their blocks can
            # go straight back to the main-stream pool.  (record_stream made the caching allocator defer their reuse
            # until it next polled the events: 37 GB peak instead of ~15 GB and hipMalloc calls inside steady-state
            # steps, i.e. occasional 40-140 ms steps.)
        self.keepalive = []
