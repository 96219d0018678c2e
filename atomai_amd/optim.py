"""FusedAdam: torch.optim.Adam semantics (defaults of the reference: trainer.py:539 lr 1e-3,
vitrainer.py:218 lr 1e-4) executed as ONE HIP launch over flat fp32 buffers (adam.hip).

It *is* a ``torch.optim.Adam`` (subclass): ``state_dict()``/pickling/param_groups keep the torch
format (``state[p] = {step, exp_avg, exp_avg_sq}``), so checkpoints written by the trainers interchange
with the reference's (atomai/trainers/trainer.py:344-358 pickles the optimizer object).

Parameters, gradients and both moments live in four flat device buffers; each parameter is re-pointed to
a 16-byte-aligned view of the flat parameter buffer, and the tape writes weight gradients straight into
views of the flat gradient buffer (``p._amx_grad``), which is also the single bucket the data-parallel
wrapper all-reduces (parallel.py).
"""
from typing import List

import torch

from . import _lib as L
from . import engine


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                         foreach=False, capturable=False)
        self._flat = None
        self.grad_scale = 1.0        # e.g. 1/world_size after a sum all-reduce

    # ------------------------------------------------------------------ flat storage
    def _params(self) -> List[torch.Tensor]:
        return [p for g in self.param_groups for p in g["params"]]

    def _layout_ok(self) -> bool:
        f = self._flat
        if f is None:
            return False
        base = f["p"].data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off in zip(f["params"], f["offsets"])) and \
            len(f["params"]) == len(self._params())

    def _flatten(self) -> None:
        ps = self._params()
        if not ps:
            raise ValueError("optimizer got an empty parameter list")
        dev = ps[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in ps):
            raise L.AmxError("FusedAdam needs fp32 parameters on one device")
        offsets, n = [], 0
        for p in ps:
            offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        old = self._flat
        flat = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
        for p, off in zip(ps, offsets):
            seg = slice(off, off + p.numel())
            flat["p"][seg].copy_(p.detach().reshape(-1))
            st = self.state.get(p, {})
            if "exp_avg" in st:                      # resumed / re-flattened state
                flat["m"][seg].copy_(st["exp_avg"].reshape(-1))
                flat["v"][seg].copy_(st["exp_avg_sq"].reshape(-1))
            p.data = flat["p"][seg].view(p.shape)
            p._amx_grad = flat["g"][seg].view(p.shape)
            if "exp_avg" in st:
                st["exp_avg"] = flat["m"][seg].view(p.shape)
                st["exp_avg_sq"] = flat["v"][seg].view(p.shape)
        flat.update(params=ps, offsets=offsets, n=n)
        self._flat = flat
        del old
        engine.bump_weight_generation()

    def flat_grad(self) -> torch.Tensor:
        """The single gradient bucket (valid after backward of a step whose grads were None before)."""
        if not self._layout_ok():
            self._flatten()
        return self._flat["g"]

    def prepare(self) -> None:
        """Flatten now (so that the first backward already writes into the flat gradient buffer)."""
        if not self._layout_ok():
            self._flatten()

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._layout_ok():
            self._flatten()
        f = self._flat
        if len(self.param_groups) != 1:
            raise L.AmxError("FusedAdam supports a single param group (as the reference trainers use)")
        grp = self.param_groups[0]
        lr, (b1, b2), eps = grp["lr"], grp["betas"], grp["eps"]
        ps, offs = f["params"], f["offsets"]
        have = [p.grad is not None for p in ps]
        if not any(have):
            return loss
        gbase = f["g"].data_ptr()
        for p, off in zip(ps, offs):
            if p.grad is not None and p.grad.data_ptr() != gbase + 4 * off:
                f["g"][off:off + p.numel()].copy_(p.grad.reshape(-1))     # device-to-device
        # lazily create torch-format state entries as views of the flat moment buffers
        for p, off in zip(ps, offs):
            st = self.state[p]
            if "exp_avg" not in st:
                seg = slice(off, off + p.numel())
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = f["m"][seg].view(p.shape)
                st["exp_avg_sq"] = f["v"][seg].view(p.shape)
        sp = L.stream_ptr(f["p"])
        steps = {int(self.state[p]["step"].item()) for p, h in zip(ps, have) if h}
        if all(have) and len(steps) == 1:               # one launch over the whole bucket (the normal case)
            t = steps.pop() + 1
            self._launch(0, f["n"], t, lr, b1, b2, eps, sp)
            for p in ps:
                self.state[p]["step"] += 1
        else:           # torch semantics: params without grad are skipped, every parameter has its OWN step count
            for p, off, h in zip(ps, offs, have):
                if not h:
                    continue
                t = int(self.state[p]["step"].item()) + 1
                self._launch(off, (p.numel() + 3) // 4 * 4, t, lr, b1, b2, eps, sp)
                self.state[p]["step"] += 1
        engine.bump_weight_generation()
        self._release_grad_views()
        return loss

    def _launch(self, off, n, t, lr, b1, b2, eps, sp):
        f = self._flat
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        seg = slice(off, off + n)
        L.call("amx_adam_flat", L.ptr(f["p"][seg]), L.ptr(f["g"][seg]), L.ptr(f["m"][seg]),
               L.ptr(f["v"][seg]), n, lr, b1, b2, eps, bc1, bc2, float(self.grad_scale), sp)

    def as_torch_adam(self) -> torch.optim.Adam:
        """Plain ``torch.optim.Adam`` over the same parameters with a copy of the current state: what the
        trainers pickle into checkpoints so that they load without this package (reference format:
        atomai/trainers/trainer.py:344-358)."""
        grp = self.param_groups[0]
        opt = torch.optim.Adam(self._params(), lr=grp["lr"], betas=grp["betas"], eps=grp["eps"])
        opt.load_state_dict(self.state_dict())
        return opt

    def _release_grad_views(self) -> None:
        for p in self._params():
            p._amx_grad_busy = False

    def zero_grad(self, set_to_none: bool = True):
        # gradients are rewritten (not accumulated) by the next backward when they are None
        self._release_grad_views()
        return super().zero_grad(set_to_none=set_to_none)
