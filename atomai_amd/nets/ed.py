"""VAE encoder / decoder networks (reference: atomai/nets/ed.py:292-343, 530-687, 725-790).

* ``fcEncoderNet`` / ``fcDecoderNet`` are dense layers on (B x features) matrices: their ``nn.Linear`` children are
  parameter containers only, the arithmetic (GEMM + bias + Tanh, forward and both gradients) runs on the fp32-MFMA
  GEMM of csrc/linear.hip (``nets/_linear.py``).
* ``rDecoderNet`` — the per-pixel spatial decoder where the step spends its time — runs the fused HIP
  kernels of csrc/rdecoder.hip (all hidden activations stay in LDS, forward and backward).
Module trees / state-dict keys / RNG-order initialisation are those of the reference.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ._linear import linear, run_dense


class fcEncoderNet(nn.Module):
    """flatten -> [Linear -> Tanh] x num_layers -> (fc11, fc12) (ed.py:292-343)."""

    def __init__(self, in_dim: Tuple[int], latent_dim: int = 2, num_layers: int = 2, hidden_dim: int = 32,
                 **kwargs: bool) -> None:
        super().__init__()
        dense = []
        for i in range(num_layers):
            input_dim = int(np.prod(in_dim)) if i == 0 else hidden_dim
            dense.extend([nn.Linear(input_dim, hidden_dim), nn.Tanh()])
        self.dense = nn.Sequential(*dense)
        self.reshape_ = hidden_dim
        self.fc11 = nn.Linear(self.reshape_, latent_dim)
        self.fc12 = nn.Linear(self.reshape_, latent_dim)
        self._out = nn.Softplus() if kwargs.get("softplus_out") else lambda x: x

    def forward(self, x: torch.Tensor):
        x = x.reshape(-1, int(np.prod(x.size()[1:])))
        x = run_dense(self.dense, x).reshape(-1, self.reshape_)
        return linear(x, self.fc11.weight, self.fc11.bias), self._out(linear(x, self.fc12.weight, self.fc12.bias))


class fcDecoderNet(nn.Module):
    """[Linear -> Tanh] x num_layers -> Linear(hidden, prod(out_dim)) (ed.py:530-580)."""

    def __init__(self, out_dim: Tuple[int], latent_dim: int, num_layers: int = 2, hidden_dim: int = 32) -> None:
        super().__init__()
        if len(out_dim) not in (1, 2, 3):
            raise ValueError("The output dimensions must be (length,) for 1D data and "
                             "(height, width) or (height, width, channel) for 2D data")
        c = out_dim[-1] if len(out_dim) > 2 else 1
        decoder = []
        for i in range(num_layers):
            hidden_dim_ = latent_dim if i == 0 else hidden_dim
            decoder.extend([nn.Linear(hidden_dim_, hidden_dim), nn.Tanh()])
        self.decoder = nn.Sequential(*decoder)
        self.out = nn.Linear(hidden_dim, int(np.prod(out_dim)))
        self.out_dim = (c, *out_dim[:2])

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        h = linear(run_dense(self.decoder, z), self.out.weight, self.out.bias).reshape(-1, *self.out_dim)
        return h.squeeze(1) if h.size(1) == 1 else h.permute(0, 2, 3, 1)


class coord_latent(nn.Module):
    """Parameter container of the decoder's first layer: Linear(2, out) on coordinates + bias-free
    Linear(latent, out) on z (ed.py:645-687).  Its arithmetic is fused into the rDecoderNet kernels."""

    def __init__(self, latent_dim: int, out_dim: int, activation: bool = False) -> None:
        super().__init__()
        self.fc_coord = nn.Linear(2, out_dim)
        self.fc_latent = nn.Linear(latent_dim, out_dim, bias=False)
        self.activation = nn.Tanh() if activation else None


import os as _os
RDEC_SAVE = [_os.environ.get("AMX_RDEC_SAVE", "1") != "0"]     # keep hidden activations for backward (see _RDecoderFn)
RDEC_SAVE_MAX_BYTES = 32 << 30                                   # beyond this the backward recomputes instead

_RDEC_WIDTHS = (32, 64, 128)          # hidden widths the fused kernels are instantiated for (rdecoder.hip)


def _zero_pad(t: torch.Tensor, shape) -> torch.Tensor:
    """Zero-extends ``t`` to ``shape`` (leading corner).  Hidden units added this way have zero weights and
    biases on both sides, hence stay exactly 0 through tanh / skip connections and change no result."""
    t = t.detach()
    if tuple(t.shape) == tuple(shape):
        return t.contiguous()
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out[tuple(slice(0, d) for d in t.shape)] = t
    return out


class _RDecoderFn(torch.autograd.Function):
    """Fused spatial decoder.  ``theta is None``: ``x_coord`` is the explicit (B, n, 2) coordinate tensor of the
    reference's forward; otherwise ``x_coord`` is the shared (n, 2) grid and ``theta`` = (B, 3) (phi, dx, dy):
    the rotation / translation of atomai/utils/coords.py:57-83 happens inside the kernel and the backward returns
    d theta instead of a (B, n, 2) coordinate gradient."""

    @staticmethod
    def forward(ctx, net, x_coord, theta, z, *params):
        fused = theta is not None
        B = z.shape[0]
        n = x_coord.shape[0] if fused else x_coord.shape[1]
        hid0, NL, Ldim, C = net.hidden_dim, net.num_layers, z.shape[1], net.channels
        hid = next(w for w in _RDEC_WIDTHS if w >= hid0)
        Wc, bc, Wz = params[0], params[1], params[2]
        Ws, bs = params[3:3 + 2 * NL:2], params[4:4 + 2 * NL:2]
        Wo, bo = params[3 + 2 * NL], params[4 + 2 * NL]
        W = _zero_pad(torch.stack([w.detach() for w in Ws]), (NL, hid, hid))
        b = _zero_pad(torch.stack([v.detach() for v in bs]), (NL, hid))
        Wc, bc, Wz = _zero_pad(Wc, (hid, 2)), _zero_pad(bc, (hid,)), _zero_pad(Wz, (hid, Ldim))
        Wo = _zero_pad(Wo, (C, hid))
        coords = x_coord.detach().contiguous()
        th = theta.detach().contiguous() if fused else None
        zz = z.detach().contiguous()
        xrec = torch.empty(B, n, C, dtype=torch.float32, device=coords.device)
        sp = L.stream_ptr(coords)
        # training: keep the hidden activations for backward (HBM round trip instead of recomputing them there) when the
        # buffer is affordable; AMX_RDEC_SAVE=0 restores the recompute-only pair
        hsave = None
        if any(ctx.needs_input_grad) and RDEC_SAVE[0]:
            nfl = L.load().amx_rdecoder_hsave_floats(B, n, hid, NL)
            if 0 < nfl * 4 <= RDEC_SAVE_MAX_BYTES:
                hsave = torch.empty(nfl, dtype=torch.float32, device=coords.device)
        L.call("amx_rdecoder_fwd_save", L.ptr(coords), L.ptr(th), L.ptr(zz), L.ptr(Wc.detach()), L.ptr(bc.detach()),
               L.ptr(Wz.detach().contiguous()), L.ptr(W), L.ptr(b), L.ptr(Wo.detach().contiguous()),
               L.ptr(bo.detach().contiguous()), L.ptr(xrec), L.ptr(hsave), B, n, Ldim, hid, NL, int(net.skip), C, sp)
        ctx.net, ctx.hid, ctx.fused = net, hid, fused
        ctx.hsave = hsave
        saved = [coords, zz, W, b] + [p.detach() for p in (Wc, bc, Wz, Wo, bo)] + ([th] if fused else [])
        ctx.save_for_backward(*saved)
        return xrec

    @staticmethod
    def backward(ctx, dxrec):
        net = ctx.net
        coords, zz, W, b, Wc, bc, Wz, Wo, bo = ctx.saved_tensors[:9]
        th = ctx.saved_tensors[9] if ctx.fused else None
        B = zz.shape[0]
        n = coords.shape[0] if ctx.fused else coords.shape[1]
        hid, hid0, NL, Ldim, C = ctx.hid, net.hidden_dim, net.num_layers, zz.shape[1], net.channels
        dev = coords.device
        Wt = W.transpose(1, 2).contiguous()
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dcoords = None if ctx.fused else e(B, n, 2)
        dtheta = e(B, 3) if ctx.fused else None
        dz = e(B, Ldim)
        pW, pb, pWo, pbo = e(B, NL * hid * hid), e(B, NL * hid), e(B, C * hid), e(B, C)
        pWc, pbc, pWz = e(B, hid * 2), e(B, hid), e(B, hid * Ldim)
        sp = L.stream_ptr(coords)
        hsave, ctx.hsave = ctx.hsave, None
        L.call("amx_rdecoder_bwd_saved", L.ptr(coords), L.ptr(th), L.ptr(zz), L.ptr(Wc), L.ptr(bc),
               L.ptr(Wz.contiguous()), L.ptr(W), L.ptr(Wt), L.ptr(b), L.ptr(Wo.contiguous()), L.ptr(bo.contiguous()),
               L.ptr(dxrec.contiguous()), L.ptr(hsave), L.ptr(dcoords), L.ptr(dtheta), L.ptr(dz), L.ptr(pW), L.ptr(pb),
               L.ptr(pWo), L.ptr(pbo), L.ptr(pWc), L.ptr(pbc), L.ptr(pWz), B, n, Ldim, hid, NL, int(net.skip), C, sp)
        del hsave

        # column sums of the seven per-sample partial tensors in TWO launches (amx_reduce_rows_segments), written straight
        # into the optimizer's flat gradient bucket where the (unpadded) parameter shape allows it
        import ctypes
        from ._linear import _grad_target
        plist = net._params()                                   # Wc, bc, Wz, (W_l, b_l)*, Wo, bo
        exact = hid == hid0
        def target(param, shape):
            t = _grad_target(param, shape, coords) if exact else None
            return t if t is not None else e(*shape)
        # one segment per parameter: (partial tensor, first column, columns, row stride) -> gradient tensor
        segs = [(pWc, 0, hid * 2, plist[0], (hid, 2)), (pbc, 0, hid, plist[1], (hid,)), (pWz, 0, hid * Ldim, plist[2], (hid, Ldim))]
        for l in range(NL):
            segs += [(pW, l * hid * hid, hid * hid, plist[3 + 2 * l], (hid, hid)), (pb, l * hid, hid, plist[4 + 2 * l], (hid,))]
        segs += [(pWo, 0, C * hid, plist[3 + 2 * NL], (C, hid)), (pbo, 0, C, plist[4 + 2 * NL], (C,))]
        outs = [target(prm, shp) for _, _, _, prm, shp in segs]
        cols = [c for _, _, c, _, _ in segs]
        nch = 32 if B > 64 else 1
        tmp = e(nch * sum(cols))
        n_ = len(segs)
        PP, LL = ctypes.c_void_p * n_, ctypes.c_long * n_
        L.call("amx_reduce_rows_segments", PP(*[t.data_ptr() + 4 * off for t, off, _, _, _ in segs]), LL(*cols),
               LL(*[t.shape[1] for t, _, _, _, _ in segs]), PP(*[t.data_ptr() for t in outs]), n_, B, nch, L.ptr(tmp), sp)
        cut = {1: lambda t: t[:hid0], 2: lambda t: t[:hid0, :hid0]}
        grads = []
        for (_, _, _, prm, shp), o in zip(segs, outs):
            o = o.view(*shp)
            if not exact:                                       # padded kernels: cut back to the parameter's shape
                if prm is plist[0] or prm is plist[2]:
                    o = o[:hid0]
                elif prm is plist[3 + 2 * NL]:
                    o = o[:, :hid0]
                elif prm is plist[4 + 2 * NL]:
                    pass
                else:
                    o = cut[len(shp)](o)
            grads.append(o)
        del outs, segs
        return (None, dcoords, dtheta, dz) + tuple(grads)


_RDEC_MAX_LATENT = 32          # MAXL of csrc/rdecoder.hip (content latents + one-hot classes)
_RDEC_MAX_LAYERS = 5
_RDEC_MAX_CHANNELS = 4


class rDecoderNet(nn.Module):
    """Spatial decoder with (optional) skip connections (ed.py:583-642) on the fused HIP kernels.

    The fused kernels cover hidden_dim <= 128 (one wave owns 16 hidden units and keeps its slice of every weight
    matrix and weight-gradient accumulator in registers; at 256 units that would be 16 waves x ~330 registers), 1-5
    hidden layers (LDS holds one activation image per layer in the backward pass) and 1-4 output channels — every
    configuration the reference's defaults and tests use.  Anything larger runs LAYER BY LAYER on the fp32-MFMA GEMM
    (csrc/linear.hip) with the activations in HBM, i.e. with the reference's own dataflow (``_forward_layered``)."""

    def __init__(self, out_dim: Tuple[int], latent_dim: int, num_layers: int, hidden_dim: int,
                 skip: bool = False) -> None:
        super().__init__()
        if len(out_dim) == 2:
            c = 1
            self.reshape_ = (out_dim[0], out_dim[1])
        else:
            c = out_dim[-1]
            self.reshape_ = (out_dim[0], out_dim[1], c)
        self.skip = skip
        self.coord_latent = coord_latent(latent_dim, hidden_dim, not skip)
        fc_decoder = []
        for i in range(num_layers):
            fc_decoder.extend([nn.Linear(hidden_dim, hidden_dim), nn.Tanh()])
        self.fc_decoder = nn.Sequential(*fc_decoder)
        self.out = nn.Linear(hidden_dim, c)
        self.hidden_dim, self.num_layers, self.channels = hidden_dim, num_layers, c

    def _fused(self) -> bool:
        return (1 <= self.channels <= _RDEC_MAX_CHANNELS and self.hidden_dim <= _RDEC_WIDTHS[-1]
                and 1 <= self.num_layers <= _RDEC_MAX_LAYERS
                and 1 <= self.coord_latent.fc_latent.in_features <= _RDEC_MAX_LATENT)

    def _forward_layered(self, x_coord: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """The reference's forward (ed.py:626-642, 672-687) layer by layer: every Linear (+ Tanh) is one launch of the
        MFMA GEMM with its own autograd (nets/_linear.py); the (B*n, hidden) activations live in HBM."""
        B, n = x_coord.shape[:2]
        cl = self.coord_latent
        h_x = linear(x_coord.reshape(B * n, -1), cl.fc_coord.weight, cl.fc_coord.bias).reshape(B, n, -1)
        h = (h_x + linear(z, cl.fc_latent.weight, None).unsqueeze(1)).reshape(B * n, -1)
        if cl.activation is not None:
            h = torch.tanh(h)
        residual = h
        for m in self.fc_decoder:
            if isinstance(m, nn.Linear):
                h = linear(h, m.weight, m.bias, "tanh")
                if self.skip:
                    h = h + residual
        return linear(h, self.out.weight, self.out.bias).reshape(B, *self.reshape_)

    def _params(self):
        params = [self.coord_latent.fc_coord.weight, self.coord_latent.fc_coord.bias,
                  self.coord_latent.fc_latent.weight]
        for m in self.fc_decoder:
            if isinstance(m, nn.Linear):
                params += [m.weight, m.bias]
        return params + [self.out.weight, self.out.bias]

    def forward(self, x_coord: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """The reference's signature: explicit (B, n, 2) coordinates."""
        if not self._fused():
            return self._forward_layered(x_coord, z)
        h = _RDecoderFn.apply(self, x_coord, None, z, *self._params())
        return h.reshape(x_coord.size(0), *self.reshape_)

    def forward_grid(self, grid: torch.Tensor, theta: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """Decodes at ``transform_coordinates(grid, theta[:, 0], theta[:, 1:3])`` without materialising the (B, n, 2)
        coordinates: ``grid`` is the shared (n, 2) ``imcoordgrid``, ``theta`` = (B, 3) (angle, dx, dy)."""
        if not self._fused():
            from ..utils.coords import transform_coordinates
            coords = transform_coordinates(grid.expand(z.size(0), *grid.shape), theta[:, 0], theta[:, None, 1:3])
            return self._forward_layered(coords.contiguous(), z)
        h = _RDecoderFn.apply(self, grid, theta, z, *self._params())
        return h.reshape(z.size(0), *self.reshape_)


class convEncoderNet(nn.Module):
    """Convolutional inference network (reference: atomai/nets/ed.py:231-289).

    ``conv`` is the HIP ConvBlock (``num_layers`` x [Conv2d 3x3 -> LeakyReLU(0.1)], no BatchNorm), followed
    by two Linear heads over the flattened (hidden_dim*H*W, NCHW order) feature map.  Inputs are (B,H,W) or
    channel-last (B,H,W,C), as in the reference.
    """

    def __init__(self, in_dim: Tuple[int], latent_dim: int = 2, num_layers: int = 2, hidden_dim: int = 32,
                 **kwargs) -> None:
        super().__init__()
        if len(in_dim) not in (1, 2, 3):
            raise ValueError("The input dimensions must be (length,) for 1D data and "
                             "(height, width) or (height, width, channel) for 2D data")
        if len(in_dim) == 1:
            raise NotImplementedError("1-D (spectral) encoders are outside the MI355X hot path")
        from .blocks import ConvBlock
        channels = in_dim[-1] if len(in_dim) > 2 else 1
        self.conv = ConvBlock(2, num_layers, channels, hidden_dim, lrelu_a=kwargs.get("lrelu_a", 0.1))
        self.reshape_ = int(hidden_dim * np.prod(in_dim[:2]))
        self.fc11 = nn.Linear(self.reshape_, latent_dim)
        self.fc12 = nn.Linear(self.reshape_, latent_dim)
        self._out = nn.Softplus() if kwargs.get("softplus_out") else (lambda t: t)

    def forward(self, x: torch.Tensor):
        x = x.unsqueeze(1) if x.ndim in (2, 3) else x.permute(0, -1, 1, 2)
        feats = self.conv(x.contiguous()).reshape(-1, self.reshape_)
        return (linear(feats, self.fc11.weight, self.fc11.bias),
                self._out(linear(feats, self.fc12.weight, self.fc12.bias)))


class convDecoderNet(nn.Module):
    """Convolutional decoder of the plain VAE (reference: atomai/nets/ed.py:471-527): bias-free Linear(latent ->
    hidden*H*W) on the MFMA GEMM -> reshape (B, hidden, H, W) -> ConvBlock(num_layers x [3x3 conv -> LeakyReLU(0.1)])
    -> 1x1 conv to the image channels, the two convolution stages in ONE tape of the HIP engine."""

    def __init__(self, out_dim: Tuple[int], latent_dim: int, num_layers: int = 2, hidden_dim: int = 32,
                 **kwargs: float) -> None:
        super().__init__()
        if len(out_dim) not in (1, 2, 3):
            raise ValueError("The output dimensions must be (length,) for 1D data and "
                             "(height, width) or (height, width, channel) for 2D data")
        if len(out_dim) == 1:
            raise NotImplementedError("1-D (spectral) decoders are outside the MI355X hot path")
        from .blocks import ConvBlock
        c = out_dim[-1] if len(out_dim) > 2 else 1
        self.fc_linear = nn.Linear(latent_dim, int(hidden_dim * np.prod(out_dim[:2])), bias=False)
        self.reshape_ = (hidden_dim, *out_dim[:2])
        self.decoder = ConvBlock(2, num_layers, hidden_dim, hidden_dim, lrelu_a=kwargs.get("lrelu_a", 0.1))
        self.conv_1x1 = nn.Conv2d(hidden_dim, c, 1, 1, 0)
        self.out_dim = (c, *out_dim[:2])

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        from ._function import run_tape
        h0 = linear(z, self.fc_linear.weight, None).reshape(-1, *self.reshape_)

        def build(tape, xin):
            node = tape.input(xin)
            act = self.decoder._emit(tape, [node.out])
            return node, tape.output(tape.conv([act], self.conv_1x1, None, 1.0))
        params = list(self.decoder.parameters()) + list(self.conv_1x1.parameters())
        h = run_tape(build, h0.contiguous(), params, self.training).reshape(-1, *self.out_dim)
        return h.squeeze(1) if h.size(1) == 1 else h.permute(0, 2, 3, 1)


def init_VAE_nets(in_dim: Tuple[int], latent_dim: int, coord: int = 0, discrete_dim: Optional[List] = None,
                  nb_classes: int = 0, **kwargs):
    """Encoder / decoder factory + metadict with the reference's keys (ed.py:725-790)."""
    conv_e = kwargs.get("conv_encoder", False)
    conv_d = kwargs.get("conv_decoder", False) if not coord else None
    numlayers_e = kwargs.get("numlayers_encoder", 2)
    numlayers_d = kwargs.get("numlayers_decoder", 2)
    numhidden_e = kwargs.get("numhidden_encoder", 128)
    numhidden_d = kwargs.get("numhidden_decoder", 128)
    skip = kwargs.get("skip", False)
    sigmoid_out = kwargs.get("sigmoid_out", False)
    softplus_out = kwargs.get("softplus_out")
    if discrete_dim:
        raise NotImplementedError("joint (discrete) VAEs are outside the MI355X hot path of this build")
    if not coord:
        dnet = convDecoderNet if conv_d else fcDecoderNet
        decoder_net = dnet(in_dim, latent_dim + nb_classes, numlayers_d, numhidden_d)
    else:
        decoder_net = rDecoderNet(in_dim, latent_dim + nb_classes, numlayers_d, numhidden_d, skip)
    enc_cls = convEncoderNet if conv_e else fcEncoderNet
    encoder_net = enc_cls(in_dim, latent_dim + coord, numlayers_e, numhidden_e, softplus_out=softplus_out)
    meta_state_dict = {"model_type": "vae", "in_dim": in_dim, "latent_dim": latent_dim, "coord": coord,
                       "conv_encoder": conv_e, "numlayers_encoder": numlayers_e,
                       "numlayers_decoder": numlayers_d, "numhidden_encoder": numhidden_e,
                       "numhidden_decoder": numhidden_d, "skip": skip, "nb_classes": nb_classes,
                       "discrete_dim": discrete_dim, "sigmoid_out": sigmoid_out, "softplus_out": softplus_out}
    if not coord:
        meta_state_dict["conv_decoder"] = conv_d
    return encoder_net, decoder_net, meta_state_dict
