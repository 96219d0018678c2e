"""Drop-in Unet / dilnet / init_fcnn_model (reference: atomai/nets/fcnn.py:18-226, 379-442)."""
from typing import List, Type, Union

import torch
import torch.nn as nn

from .blocks import ConvBlock, DilatedBlock, ResModule, UpsampleBlock
from ._function import run_tape


_warned_modular = False


class _HipNet(nn.Module):
    def _build(self, tape, x):
        raise NotImplementedError

    def _modular(self, x):
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # Utilities such as get_downsample_factor hook every top-level child and need it invoked via
        # __call__ (atomai/utils/nn.py:215-228): honour that with the block-by-block path.
        if any(len(m._forward_hooks) or len(m._forward_pre_hooks) for m in self.children()):
            global _warned_modular
            if not _warned_modular:
                _warned_modular = True
                import warnings
                warnings.warn("a forward hook is attached to a block of this network: the forward pass runs block by block "
                              "(every block, pooling and the final 1x1 convolution on their HIP kernels; torch.cat only moves "
                              "data) instead of the fused single-tape path, so that the hook sees its block's input and "
                              "output", RuntimeWarning)
            return self._modular(x)
        return run_tape(self._build, x, list(self.parameters()), self.training)

    # ---- the trainers' fused step: prob = net(x); loss = CrossEntropyLoss()(prob, y) as ONE tape whose last node is the
    # head + loss + their backward in a single pass over the last activation (engine.PxLossNode)
    _loss_target = None
    _loss_fused = False

    def _px(self, tape, act, conv, px_mode: int):
        from ..engine import px_loss_fusable
        tgt = self._loss_target
        if tgt is not None and tape.training and px_mode == 0 and px_loss_fusable(act, conv, tgt):
            self._loss_fused = True
            return tape.px(act, conv, 0, loss_target=tgt)
        return tape.px(act, conv, px_mode)

    def forward_loss(self, x: torch.Tensor, target: torch.Tensor):
        """('loss', mean cross-entropy of net(x) against the int64 class map `target` [N][H][W] — for a one-class net the mean
        BCE-with-logits against the float mask [N][1][H][W]) when the head and the loss run as the fused node, else
        ('logits', net(x)) — the caller then applies its criterion as usual.  Training mode only."""
        if not self.training or any(len(m._forward_hooks) or len(m._forward_pre_hooks) for m in self.children()):
            return "logits", self.forward(x)
        self._loss_target, self._loss_fused = target, False
        try:
            out = run_tape(self._build, x, list(self.parameters()), True)
        finally:
            self._loss_target = None
        return ("loss" if self._loss_fused else "logits"), out


def _hip_pool(x: torch.Tensor, training: bool) -> torch.Tensor:
    """2x2 max-pooling of an NCHW tensor on the HIP pooling kernels (the glue of the block-by-block path)."""
    def build(tape, xin):
        node = tape.input(xin)
        return node, tape.output(tape.pool(node.out))
    return run_tape(build, x, [], training)


def _hip_px(x: torch.Tensor, px: nn.Conv2d, training: bool) -> torch.Tensor:
    """The final 1x1 convolution to class logits on the HIP px kernel (not nn.Conv2d -> MIOpen), invoked through the
    module's own ``__call__`` so that hooks attached to it (get_downsample_factor hooks every top-level child) still fire."""
    def build(tape, xin):
        node = tape.input(xin)
        return node, tape.px(node.out, px, 0)
    px.forward = lambda xin: run_tape(build, xin, list(px.parameters()), training)     # instance attribute, for one call
    try:
        return px(x)
    finally:
        del px.forward


class Unet(_HipNet):
    """U-Net: c1-pool-c2-pool-c3-pool-bn-up1-cat-c4-up2-cat-c5-up3-cat-c6-px (fcnn.py:18-142)."""

    def __init__(self, nb_classes: int = 1, nb_filters: int = 16, dropout: bool = False,
                 batch_norm: bool = True, upsampling_mode: str = "bilinear", with_dilation: bool = False,
                 **kwargs: List[int]) -> None:
        super().__init__()
        nbl = kwargs.get("layers", [1, 2, 2, 3])
        dilation_values = torch.arange(2, 2 * nbl[-1] + 1, 2).tolist()
        padding_values = dilation_values.copy()
        dropout_vals = [.1, .2, .1] if dropout else [0, 0, 0]
        self.c1 = ConvBlock(2, nbl[0], 1, nb_filters, batch_norm=batch_norm)
        self.c2 = ConvBlock(2, nbl[1], nb_filters, nb_filters * 2, batch_norm=batch_norm)
        self.c3 = ConvBlock(2, nbl[2], nb_filters * 2, nb_filters * 4, batch_norm=batch_norm,
                            dropout_=dropout_vals[0])
        if with_dilation:
            self.bn = DilatedBlock(2, nb_filters * 4, nb_filters * 8, dilation_values=dilation_values,
                                   padding_values=padding_values, batch_norm=batch_norm,
                                   dropout_=dropout_vals[1])
        else:
            self.bn = ConvBlock(2, nbl[3], nb_filters * 4, nb_filters * 8, batch_norm=batch_norm,
                                dropout_=dropout_vals[1])
        self.upsample_block1 = UpsampleBlock(2, nb_filters * 8, nb_filters * 4, mode=upsampling_mode)
        self.c4 = ConvBlock(2, nbl[2], nb_filters * 8, nb_filters * 4, batch_norm=batch_norm,
                            dropout_=dropout_vals[2])
        self.upsample_block2 = UpsampleBlock(2, nb_filters * 4, nb_filters * 2, mode=upsampling_mode)
        self.c5 = ConvBlock(2, nbl[1], nb_filters * 4, nb_filters * 2, batch_norm=batch_norm)
        self.upsample_block3 = UpsampleBlock(2, nb_filters * 2, nb_filters, mode=upsampling_mode)
        self.c6 = ConvBlock(2, nbl[0], nb_filters * 2, nb_filters, batch_norm=batch_norm)
        self.px = nn.Conv2d(nb_filters, nb_classes, 1, 1, 0)

    def _build(self, tape, x, px_mode: int = 0):
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise AssertionError("Unet needs H and W divisible by 8 (three 2x2 poolings); "
                                 "SegPredictor pads inputs accordingly")
        node, c1 = self.c1._emit_input(tape, x, pool_next=True)
        d1 = tape.pool(c1)
        c2 = self.c2._emit(tape, [d1])
        d2 = tape.pool(c2)
        c3 = self.c3._emit(tape, [d2])
        d3 = tape.pool(c3)
        bn = self.bn._emit(tape, [d3])
        u3 = self.upsample_block1._emit(tape, [bn])
        u3 = self.c4._emit(tape, [c3, u3])
        u2 = self.upsample_block2._emit(tape, [u3])
        u2 = self.c5._emit(tape, [c2, u2])
        u1 = self.upsample_block3._emit(tape, [u2])
        u1 = self.c6._emit(tape, [c1, u1], head=(self.px, px_mode))
        return node, (u1 if hasattr(u1, "value") else self._px(tape, u1, self.px, px_mode))

    def _modular(self, x):
        c1 = self.c1(x)
        d1 = _hip_pool(c1, self.training)
        c2 = self.c2(d1)
        d2 = _hip_pool(c2, self.training)
        c3 = self.c3(d2)
        d3 = _hip_pool(c3, self.training)
        bn = self.bn(d3)
        u3 = self.c4(torch.cat([c3, self.upsample_block1(bn)], dim=1))
        u2 = self.c5(torch.cat([c2, self.upsample_block2(u3)], dim=1))
        u1 = self.c6(torch.cat([c1, self.upsample_block3(u2)], dim=1))
        return _hip_px(u1, self.px, self.training)


class dilnet(_HipNet):
    """c1-pool-at1-at2-up1-cat(c1,u1)-c2-px (fcnn.py:145-226)."""

    def __init__(self, nb_classes: int = 1, nb_filters: int = 25, dropout: bool = False,
                 batch_norm: bool = True, upsampling_mode: str = "bilinear", **kwargs: List[int]) -> None:
        super().__init__()
        nbl = kwargs.get("layers", [3, 3, 3, 3])
        dilation_values_1 = torch.arange(2, 2 * nbl[1] + 1, 2).tolist()
        padding_values_1 = dilation_values_1.copy()
        dilation_values_2 = torch.arange(2, 2 * nbl[2] + 1, 2).tolist()
        padding_values_2 = dilation_values_2.copy()
        dropout_vals = [.3, .3] if dropout else [0, 0]
        self.c1 = ConvBlock(2, nbl[0], 1, nb_filters, batch_norm=batch_norm)
        self.at1 = DilatedBlock(2, nb_filters, nb_filters * 2, dilation_values=dilation_values_1,
                                padding_values=padding_values_1, batch_norm=batch_norm,
                                dropout_=dropout_vals[0])
        self.at2 = DilatedBlock(2, nb_filters * 2, nb_filters * 2, dilation_values=dilation_values_2,
                                padding_values=padding_values_2, batch_norm=batch_norm,
                                dropout_=dropout_vals[1])
        self.up1 = UpsampleBlock(2, nb_filters * 2, nb_filters, mode=upsampling_mode)
        self.c2 = ConvBlock(2, nbl[3], nb_filters * 2, nb_filters, batch_norm=batch_norm)
        self.px = nn.Conv2d(nb_filters, nb_classes, 1, 1, 0)

    def _build(self, tape, x, px_mode: int = 0):
        if x.shape[2] % 2 or x.shape[3] % 2:
            raise AssertionError("dilnet needs even H and W (one 2x2 pooling)")
        node, c1 = self.c1._emit_input(tape, x, pool_next=True)
        d1 = tape.pool(c1)
        at1 = self.at1._emit(tape, [d1])
        at2 = self.at2._emit(tape, [at1])
        u1 = self.up1._emit(tape, [at2])
        u1 = self.c2._emit(tape, [c1, u1], head=(self.px, px_mode))
        return node, (u1 if hasattr(u1, "value") else self._px(tape, u1, self.px, px_mode))

    def _modular(self, x):
        c1 = self.c1(x)
        d1 = _hip_pool(c1, self.training)
        u1 = self.up1(self.at2(self.at1(d1)))
        return _hip_px(self.c2(torch.cat([c1, u1], dim=1)), self.px, self.training)


class ResHedNet(_HipNet):
    """Holistically-nested edge detector with residual blocks (reference: atomai/nets/fcnn.py:229-295): three
    ResModules at full / half / quarter resolution, a 1x1 conv + BatchNorm side output from each, the two coarse ones
    interpolated back to the input size, concatenated and fused by a 1x1 conv."""

    def __init__(self, nb_classes: int = 1, nb_filters: int = 64, upsampling_mode: str = "bilinear",
                 **kwargs: List[int]) -> None:
        super().__init__()
        nbl = kwargs.get("layers", [3, 4, 5])
        if upsampling_mode not in ("bilinear", "nearest"):
            raise NotImplementedError("use 'bilinear' or 'nearest' for upsampling mode")
        self.upsample = upsampling_mode
        self.net1 = ResModule(2, nbl[0], 1, nb_filters, True)
        self.net2 = nn.Sequential(nn.MaxPool2d(2, 2), ResModule(2, nbl[1], nb_filters, 2 * nb_filters, True))
        self.net3 = nn.Sequential(nn.MaxPool2d(2, 2), ResModule(2, nbl[2], 2 * nb_filters, 4 * nb_filters, True))
        self.net1score = nn.Sequential(nn.Conv2d(nb_filters, nb_classes, 1, 1, 0), nn.BatchNorm2d(nb_classes))
        self.net2score = nn.Sequential(nn.Conv2d(2 * nb_filters, nb_classes, 1, 1, 0), nn.BatchNorm2d(nb_classes))
        self.net3score = nn.Sequential(nn.Conv2d(4 * nb_filters, nb_classes, 1, 1, 0), nn.BatchNorm2d(nb_classes))
        self.out = nn.Conv2d(3 * nb_classes, nb_classes, 1, 1, 0)

    def _build(self, tape, x, px_mode: int = 0):
        h, w = x.shape[2:4]
        node = tape.input(x)
        n1 = self.net1._emit(tape, [node.out])
        n2 = self.net2[1]._emit(tape, [tape.pool(n1)])
        n3 = self.net3[1]._emit(tape, [tape.pool(n2)])
        scores = [tape.conv([n], seq[0], seq[1], 1.0)                       # conv 1x1 -> BatchNorm, no activation
                  for n, seq in ((n1, self.net1score), (n2, self.net2score), (n3, self.net3score))]
        cat = tape.resize_cat(scores, h, w, self.upsample)
        return node, self._px(tape, cat, self.out, px_mode)

    def _modular(self, x):
        # (hook path only: the side-output heads — pooling inside net2 / net3, three 1x1 convolutions on nb_classes channels,
        #  their interpolation and the fusing 1x1 convolution — are stock torch modules here; the residual modules, which
        #  hold the work, run on their HIP kernels.  The product path is _build.)
        import torch.nn.functional as F
        h, w = x.shape[2:4]
        n1 = self.net1(x)
        n2 = self.net2(n1)
        n3 = self.net3(n2)
        s1, s2, s3 = self.net1score(n1), self.net2score(n2), self.net3score(n3)
        s2 = F.interpolate(s2, size=(h, w), mode=self.upsample)
        s3 = F.interpolate(s3, size=(h, w), mode=self.upsample)
        return self.out(torch.cat([s1, s2, s3], 1))


class SegResNet(_HipNet):
    """SegNet-like net with residual blocks: c1-pool-c2(res)-pool-bn(res)-up1-cat-c3(res)-up2-cat-c4-px
    (reference: atomai/nets/fcnn.py:297-376)."""

    def __init__(self, nb_classes: int = 1, nb_filters: int = 32, batch_norm: bool = True,
                 upsampling_mode: str = "bilinear", **kwargs: List[int]) -> None:
        super().__init__()
        nbl = kwargs.get("layers", [2, 2, 2])
        self.c1 = ConvBlock(2, 1, 1, nb_filters, batch_norm=batch_norm)
        self.c2 = ResModule(2, nbl[0], nb_filters, nb_filters * 2, batch_norm=batch_norm)
        self.bn = ResModule(2, nbl[1], nb_filters * 2, nb_filters * 4, batch_norm=batch_norm)
        self.upsample_block1 = UpsampleBlock(2, nb_filters * 4, nb_filters * 2, 2, upsampling_mode)
        self.c3 = ResModule(2, nbl[2], nb_filters * 4, nb_filters * 2, batch_norm=batch_norm)
        self.upsample_block2 = UpsampleBlock(2, nb_filters * 2, nb_filters, 2, upsampling_mode)
        self.c4 = ConvBlock(2, 1, nb_filters * 2, nb_filters, batch_norm=batch_norm)
        self.px = nn.Conv2d(nb_filters, nb_classes, 1, 1, 0)

    def _build(self, tape, x, px_mode: int = 0):
        if x.shape[2] % 4 or x.shape[3] % 4:
            raise AssertionError("SegResNet needs H and W divisible by 4 (two 2x2 poolings); "
                                 "SegPredictor pads inputs accordingly")
        node, c1 = self.c1._emit_input(tape, x)
        d1 = tape.pool(c1)
        c2 = self.c2._emit(tape, [d1])
        d2 = tape.pool(c2)
        bn = self.bn._emit(tape, [d2])
        u2 = self.upsample_block1._emit(tape, [bn])
        u2 = self.c3._emit(tape, [c2, u2])
        u1 = self.upsample_block2._emit(tape, [u2])
        u1 = self.c4._emit(tape, [c1, u1])
        return node, self._px(tape, u1, self.px, px_mode)

    def _modular(self, x):
        c1 = self.c1(x)
        c2 = self.c2(_hip_pool(c1, self.training))
        bn = self.bn(_hip_pool(c2, self.training))
        u2 = self.c3(torch.cat([c2, self.upsample_block1(bn)], dim=1))
        u1 = self.c4(torch.cat([c1, self.upsample_block2(u2)], dim=1))
        return _hip_px(u1, self.px, self.training)


def init_fcnn_model(model: Union[Type[nn.Module], str], nb_classes: int, **kwargs):
    """Factory + meta_state_dict, same keys as the reference (fcnn.py:379-442)."""
    if not isinstance(model, str) and hasattr(model, "state_dict"):
        meta_state_dict = {'model_type': 'Seg', model: 'custom', 'nb_classes': nb_classes}
        return model, meta_state_dict
    batch_norm = kwargs.get('batch_norm', True)
    dropout = kwargs.get('dropout', False)
    upsampling = kwargs.get('upsampling', kwargs.get('upsampling_mode', "bilinear"))
    meta_state_dict = {'model_type': 'seg', 'model': model, 'nb_classes': nb_classes,
                       'batch_norm': batch_norm, 'dropout': dropout, 'upsampling': upsampling}
    if isinstance(model, str) and model == 'Unet':
        with_dilation = kwargs.get('with_dilation', False)
        nb_filters = kwargs.get('nb_filters', 16)
        layers = kwargs.get("layers", [1, 2, 2, 3])
        net = Unet(nb_classes, nb_filters, dropout, batch_norm, upsampling, with_dilation, layers=layers)
        meta_state_dict["with_dilation"] = with_dilation
    elif isinstance(model, str) and model == 'dilnet':
        nb_filters = kwargs.get('nb_filters', 25)
        layers = kwargs.get("layers", [1, 3, 3, 1])
        net = dilnet(nb_classes, nb_filters, dropout, batch_norm, upsampling, layers=layers)
    elif isinstance(model, str) and model == 'SegResNet':
        nb_filters = kwargs.get('nb_filters', 32)
        layers = kwargs.get("layers", [2, 2, 2])
        net = SegResNet(nb_classes, nb_filters, batch_norm, upsampling, layers=layers)
    elif isinstance(model, str) and model == 'ResHedNet':
        nb_filters = kwargs.get('nb_filters', 64)
        layers = kwargs.get("layers", [3, 4, 5])
        net = ResHedNet(nb_classes, nb_filters, upsampling, layers=layers)
    else:
        raise NotImplementedError(
            "Currently implemented models are 'Unet', 'dilnet', SegResNet', and 'ResHedNet'")
    if model in ["ResHedNet", "SegResNet"]:
        meta_state_dict["dropout"] = None
    meta_state_dict["nb_filters"] = nb_filters
    meta_state_dict["layers"] = layers
    return net, meta_state_dict


def predict_proba(net: _HipNet, x: torch.Tensor, input_norm=None) -> torch.Tensor:
    """Eval-mode forward returning class probabilities in NHWC — sigmoid (1 class) / softmax fused into
    the px kernel together with the NCHW->NHWC permute of SegPredictor.forward_
    (atomai/predictors/predictor.py:219-229).  ``input_norm`` = (min, ptp): the net sees (x - min) / ptp, applied by
    the first-layer kernel while it loads (nets whose first layer is not the single-channel 3x3 kernel get a separate
    normalisation pass)."""
    from ..engine import Tape
    from .. import _lib as L
    assert not net.training
    with torch.no_grad():
        tape = Tape(False, False)
        tape.input_norm = None if input_norm is None else (float(input_norm[0]), float(input_norm[1]))
        if input_norm is not None and not _fuses_input_norm(net, x):
            raw = x.contiguous()
            x = torch.empty_like(raw)
            L.call("amx_sub_div", L.ptr(raw), L.ptr(x), raw.numel(), tape.input_norm[0], tape.input_norm[1],
                   L.stream_ptr(raw))
            tape.input_norm = None
        out = net._build(tape, x, px_mode=1)[1].value
        assert tape.input_norm is None or tape.input_norm_used
        return out


def _fuses_input_norm(net, x) -> bool:
    """True if the net's first layer runs on the single-channel first-layer kernel (ConvBlock._emit_input)."""
    c1 = getattr(net, "c1", None)
    conv0 = c1.block[0] if isinstance(c1, ConvBlock) else None
    return (isinstance(net, (Unet, dilnet)) and conv0 is not None and x.shape[1] == 1 and conv0.in_channels == 1
            and tuple(conv0.kernel_size) == (3, 3))
