"""Drop-in ConvBlock / UpsampleBlock / DilatedBlock (reference: atomai/nets/blocks.py:17-132, 257-329).

Same constructor signatures, same module tree (``self.block`` / ``self.conv`` / ``self.atrous_module``
are nn.Sequential / nn.Conv2d holding nn.Conv2d / nn.LeakyReLU / nn.BatchNorm2d children), hence the
same ``state_dict()`` keys, shapes and RNG-order initialisation as the reference.  The children are
parameter containers only: ``forward`` never calls them — it emits HIP kernels through engine.Tape.
"""
from typing import List, Sequence, Tuple, Union

import torch
import torch.nn as nn

from ..engine import Act, Tape
from ._function import run_tape


def _layers(seq: nn.Sequential):
    """Groups an nn.Sequential [conv, (dropout), lrelu, (bn)]* into (conv, slope, bn, dropout) tuples."""
    out, cur = [], None
    for m in seq:
        if isinstance(m, (nn.Conv2d,)):
            if cur:
                out.append(cur)
            cur = [m, 1.0, None, None]
        elif isinstance(m, nn.LeakyReLU):
            cur[1] = float(m.negative_slope)
        elif isinstance(m, nn.BatchNorm2d):
            cur[2] = m
        elif isinstance(m, nn.Dropout):
            cur[3] = m
        else:
            raise NotImplementedError(f"layer {type(m).__name__} is not on the HIP hot path")
    if cur:
        out.append(cur)
    return out


def _drop_p(drop) -> float:
    return float(drop.p) if drop is not None else 0.0


class _HipBlock(nn.Module):
    """Common forward: NCHW tensor -> tape -> NCHW tensor (module-boundary behaviour of the reference)."""

    def _emit(self, tape: Tape, srcs: Sequence[Act]) -> Act:
        raise NotImplementedError

    def _emit_input(self, tape: Tape, x: torch.Tensor):
        """Returns (input node or None, Act) for an NCHW tensor entering this block."""
        node = tape.input(x)
        return node, self._emit(tape, [node.out])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim != 4:
            raise AssertionError("expected a (N, C, H, W) tensor")

        def build(tape, xin):
            node, act = self._emit_input(tape, xin)
            return node, tape.output(act)
        return run_tape(build, x, list(self.parameters()), self.training)


class ConvBlock(_HipBlock):
    """nb_layers x [Conv2d k3 s1 p1 -> (Dropout) -> LeakyReLU -> (BatchNorm2d)]  (blocks.py:17-83)."""

    def __init__(self, ndim: int, nb_layers: int, input_channels: int, output_channels: int,
                 kernel_size: Union[Tuple[int], int] = 3, stride: Union[Tuple[int], int] = 1,
                 padding: Union[Tuple[int], int] = 1, batch_norm: bool = False, lrelu_a: float = 0.01,
                 dropout_: float = 0) -> None:
        super().__init__()
        if not 0 < ndim < 3:
            raise AssertionError("ndim must be equal to 1 or 2")
        if ndim == 1:
            raise NotImplementedError("1-D ConvBlock (ImSpec family) is outside the MI355X hot path")
        block = []
        for idx in range(nb_layers):
            input_channels = output_channels if idx > 0 else input_channels
            block.append(nn.Conv2d(input_channels, output_channels, kernel_size=kernel_size,
                                   stride=stride, padding=padding))
            if dropout_ > 0:
                block.append(nn.Dropout(dropout_))
            block.append(nn.LeakyReLU(negative_slope=lrelu_a))
            if batch_norm:
                block.append(nn.BatchNorm2d(output_channels))
        self.block = nn.Sequential(*block)

    def _emit(self, tape, srcs, head=None):
        """head = (px conv, mode): the net's classification head, evaluated in the epilogue of this block's last layer
        when that is possible (eval mode; engine.head_fusable) — then the fused node is returned instead of an Act."""
        from ..engine import head_fusable
        layers = _layers(self.block)
        for i, (conv, slope, bn, drop) in enumerate(layers):
            if head is not None and i == len(layers) - 1 and head_fusable(tape, srcs, conv, head[0]):
                return tape.conv_head(srcs, conv, bn, slope, head[0], head[1])
            srcs = [tape.conv(srcs, conv, bn, slope, drop_p=_drop_p(drop))]
        return srcs[0]

    def _emit_input(self, tape, x, pool_next: bool = False):
        """pool_next: the caller max-pools this block's output next (lets a one-layer block's first-layer kernel
        produce the pooled tensor too in eval mode)."""
        layers = _layers(self.block)
        conv0 = layers[0][0]
        fast = (x.shape[1] == 1 and conv0.in_channels == 1 and conv0.kernel_size == (3, 3)
                and not (x.requires_grad and tape.need_grad))
        if not fast:
            return super()._emit_input(tape, x)
        act = tape.conv_first(x, conv0, layers[0][2], layers[0][1], drop_p=_drop_p(layers[0][3]),
                              pool_next=pool_next and len(layers) == 1)
        for conv, slope, bn, drop in layers[1:]:
            act = tape.conv([act], conv, bn, slope, drop_p=_drop_p(drop))
        return None, act


class UpsampleBlock(_HipBlock):
    """F.interpolate(x2, bilinear|nearest) -> Conv2d 1x1 (blocks.py:86-132).  The 1x1 convolution (with
    bias) commutes exactly with the interpolation (its weights sum to 1), so it is evaluated at LOW
    resolution and the result is upsampled: 4x fewer FLOPs, the wide high-res tensor never exists."""

    def __init__(self, ndim: int, input_channels: int, output_channels: int, scale_factor: int = 2,
                 mode: str = "bilinear") -> None:
        super().__init__()
        if not any([mode == 'bilinear', mode == 'nearest']):
            raise NotImplementedError("use 'bilinear' or 'nearest' for upsampling mode")
        if not 0 < ndim < 3:
            raise AssertionError("ndim must be equal to 1 or 2")
        if ndim == 1:
            raise NotImplementedError("1-D UpsampleBlock is outside the MI355X hot path")
        if scale_factor != 2:
            raise NotImplementedError("only scale_factor=2 is on the MI355X hot path")
        self.scale_factor = scale_factor
        self.mode = mode
        self.conv = nn.Conv2d(input_channels, output_channels, kernel_size=1, stride=1, padding=0)

    def _emit(self, tape, srcs):
        from ..engine import upconv_fusable
        if len(srcs) == 1 and upconv_fusable(srcs[0], self.conv):
            return tape.upconv(srcs[0], self.conv, self.mode)          # one launch, same bits (csrc/upconv.hip)
        v = tape.conv(srcs, self.conv, None, 1.0)
        return tape.upsample(v, self.mode)


class DilatedBlock(_HipBlock):
    """Cascade of dilated 3x3 convs whose output is the sum of EVERY sub-layer output (blocks.py:257-329)."""

    def __init__(self, ndim: int, input_channels: int, output_channels: int, dilation_values: List[int],
                 padding_values: List[int], kernel_size: Union[Tuple[int], int] = 3,
                 stride: Union[Tuple[int], int] = 1, lrelu_a: float = 0.01, batch_norm: bool = False,
                 dropout_: float = 0) -> None:
        super().__init__()
        if not 0 < ndim < 3:
            raise AssertionError("ndim must be equal to 1 or 2")
        if ndim == 1:
            raise NotImplementedError("1-D DilatedBlock is outside the MI355X hot path")
        atrous_module = []
        for idx, (dil, pad) in enumerate(zip(dilation_values, padding_values)):
            input_channels = output_channels if idx > 0 else input_channels
            atrous_module.append(nn.Conv2d(input_channels, output_channels, kernel_size=kernel_size,
                                           stride=stride, padding=pad, dilation=dil, bias=True))
            if dropout_ > 0:
                atrous_module.append(nn.Dropout(dropout_))
            atrous_module.append(nn.LeakyReLU(negative_slope=lrelu_a))
            if batch_norm:
                atrous_module.append(nn.BatchNorm2d(output_channels))
        self.atrous_module = nn.Sequential(*atrous_module)

    def _emit(self, tape, srcs):
        from ..engine import dsum_fusable
        acts, slope = [], 0.01
        layers = _layers(self.atrous_module)
        # A Dropout layer is one more sub-layer whose OUTPUT is summed (blocks.py:311-312, 321-329): in eval mode it is the
        # identity (the convolution output counts twice), in training the dropped AND the un-dropped tensor are summed.
        has_drop = any(d is not None for _, _, _, d in layers)
        dropping = has_drop and tape.training and any(_drop_p(d) > 0 for _, _, _, d in layers)
        for i, (conv, slope, bn, drop) in enumerate(layers):
            if i == len(layers) - 1 and not has_drop and dsum_fusable(tape, srcs, conv, acts):
                return tape.conv_dsum(srcs[0], conv, bn, slope, acts)      # eval: the block's sum in this layer's epilogue
            a = tape.conv(srcs, conv, bn, slope, drop_p=_drop_p(drop) if dropping else 0.0, keep_unmasked=dropping)
            acts.append(a)
            srcs = [a]
        return tape.dilated_sum(acts, slope, wpre=2.0 if (has_drop and not dropping) else 1.0)


class ResBlock(_HipBlock):
    """Residual block (reference: atomai/nets/blocks.py:135-214): x = c0(x) [1x1]; out = c1(x) -> bn1 -> LeakyReLU
    -> c2 -> bn2; out += x; LeakyReLU(out).  BatchNorm sits BEFORE the activation here, so bn1's affine and the
    activation are applied by c2's loader and bn2's affine by the residual kernel."""

    def __init__(self, ndim: int, input_channels: int, output_channels: int,
                 kernel_size: Union[Tuple[int], int] = 3, stride: Union[Tuple[int], int] = 1,
                 padding: Union[Tuple[int], int] = 1, batch_norm: bool = True, lrelu_a: float = 0.01) -> None:
        super().__init__()
        if not 0 < ndim < 3:
            raise AssertionError("ndim must be equal to 1 or 2")
        if ndim == 1:
            raise NotImplementedError("1-D ResBlock (ImSpec family) is outside the MI355X hot path")
        self.lrelu_a = lrelu_a
        self.batch_norm = batch_norm
        self.c0 = nn.Conv2d(input_channels, output_channels, kernel_size=1, stride=1, padding=0)
        self.c1 = nn.Conv2d(output_channels, output_channels, kernel_size=3, stride=1, padding=1)
        self.c2 = nn.Conv2d(output_channels, output_channels, kernel_size=3, stride=1, padding=1)
        if batch_norm:
            self.bn1 = nn.BatchNorm2d(output_channels)
            self.bn2 = nn.BatchNorm2d(output_channels)

    def _emit(self, tape, srcs):
        bn1 = self.bn1 if self.batch_norm else None
        bn2 = self.bn2 if self.batch_norm else None
        x0 = tape.conv(srcs, self.c0, None, 1.0)
        h1 = tape.conv([x0], self.c1, bn1, 1.0, post_slope=self.lrelu_a)
        t2 = tape.conv([h1], self.c2, bn2, 1.0)
        return tape.res_out(t2, x0, self.lrelu_a)


class ResModule(_HipBlock):
    """``res_depth`` residual blocks in sequence (reference: atomai/nets/blocks.py:217-254)."""

    def __init__(self, ndim: int, res_depth: int, input_channels: int, output_channels: int,
                 batch_norm: bool = True, lrelu_a: float = 0.01) -> None:
        super().__init__()
        res_module = []
        for i in range(res_depth):
            input_channels = output_channels if i > 0 else input_channels
            res_module.append(ResBlock(ndim, input_channels, output_channels, lrelu_a=lrelu_a,
                                       batch_norm=batch_norm))
        self.res_module = nn.Sequential(*res_module)

    def _emit(self, tape, srcs):
        act = None
        for blk in self.res_module:
            act = blk._emit(tape, srcs)
            srcs = [act]
        return act
