"""Bridges a HIP tape (engine.Tape) into torch.autograd as ONE Function per module call."""
from typing import Callable, Sequence

import torch

from ..engine import Tape


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, build, training, x, *params):
        tape = Tape(training, True)
        in_node, out_node = build(tape, x)
        tape.end_forward()
        ctx.tape, ctx.in_node, ctx.out_node, ctx.params = tape, in_node, out_node, params
        # The returned tensor will own this Function's grad_fn, which owns ctx: keeping it reachable from ctx
        # (ctx -> out_node -> value) would close a reference cycle through the C++ autograd node, and the step's
        # last activations (~1.3 GB at the benchmark size) would then pile up until the cyclic GC runs.
        value, out_node.value = out_node.value, None
        return value

    @staticmethod
    def backward(ctx, gout):
        tape = ctx.tape
        ctx.out_node.grad_out = gout
        tape.backward()
        gx = ctx.in_node.grad_nchw if ctx.in_node is not None else None
        grads = []
        for p in ctx.params:
            hit = tape.param_grads.get(id(p))
            grads.append(hit[1].view(p.shape) if hit is not None else None)
        ctx.tape = ctx.in_node = ctx.out_node = None
        return (None, None, gx) + tuple(grads)


def run_tape(build: Callable, x: torch.Tensor, params: Sequence[torch.Tensor], training: bool):
    """build(tape, x) -> (input node or None, output node with .value/.grad_out)."""
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need:
        tape = Tape(training, False)
        value = build(tape, x)[1].value
        tape.end_forward()
        return value
    return _TapeFn.apply(build, training, x, *params)
