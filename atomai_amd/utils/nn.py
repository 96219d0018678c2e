"""Seeds, hooks and weight utilities (reference: atomai/utils/nn.py:59-81, 136-249)."""
import copy
from typing import Dict, Tuple, Type

import numpy as np
import torch
from torch.nn import BatchNorm1d, BatchNorm2d, Conv1d, Conv2d, ConvTranspose1d, ConvTranspose2d, Linear


def set_train_rng(seed: int = 1) -> None:
    """numpy + torch (+ all GPUs) seeding, as utils/nn.py:136-146."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Hook:
    """Records input/output of a module during a forward (or backward) pass (utils/nn.py:169-192)."""

    def __init__(self, module: Type[torch.nn.Module], backward: bool = False) -> None:
        reg = module.register_full_backward_hook if backward else module.register_forward_hook
        self.hook = reg(self.hook_fn)

    def hook_fn(self, module, input_, output_) -> None:
        self.input, self.output = input_, output_

    def close(self) -> None:
        self.hook.remove()


def mock_forward(model: Type[torch.nn.Module], dims: Tuple[int] = (1, 64, 64)) -> torch.Tensor:
    """Passes a dummy variable through a network on the model's device (utils/nn.py:195-209)."""
    x = torch.randn(1, *dims)
    x = x.to(next(model.parameters()).device)
    with torch.no_grad():
        return model(x)


def _hooked_shapes(model):
    """Output shapes of every top-level child for a 1x1x64x64 mock input.  Unlike the reference, the mock
    forward runs in eval mode (restored afterwards) so that it cannot disturb BatchNorm running statistics."""
    hooks = [Hook(child) for _, child in model._modules.items()]
    was_training = model.training
    model.eval()
    try:
        mock_forward(model)
        return [h.output.shape for h in hooks]
    finally:
        model.train(was_training)
        for h in hooks:
            h.close()


def get_nb_classes(model: Type[torch.nn.Module]) -> int:
    """Channel count of the last top-level child's output (utils/nn.py:212-218)."""
    return _hooked_shapes(model)[-1][1]


def get_downsample_factor(model: Type[torch.nn.Module]) -> int:
    """max/min spatial size over the top-level children's outputs (utils/nn.py:221-228)."""
    sizes = [s[-1] for s in _hooked_shapes(model)]
    return max(sizes) / min(sizes)


def weights_init(module) -> None:
    """Xavier-uniform weights, zero biases (utils/nn.py:238-242).  The values are always drawn from the CPU
    generator and copied to the parameter's device, so that a seed gives the same ensemble member on any device
    (the reference draws from whichever generator the parameter lives on: its GPU and CPU members differ)."""
    if isinstance(module, (Conv1d, Conv2d, ConvTranspose1d, ConvTranspose2d, Linear)):
        w = torch.empty(module.weight.shape, dtype=module.weight.dtype)
        torch.nn.init.xavier_uniform_(w)
        with torch.no_grad():
            module.weight.copy_(w)
            if module.bias is not None:
                module.bias.zero_()


def reset_bnorm(module) -> None:
    if isinstance(module, (BatchNorm1d, BatchNorm2d)):
        module.reset_running_stats()
        module.reset_parameters()


def average_weights(ensemble: Dict[int, Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Averages every non-BN-statistic tensor over an ensemble of state dicts (utils/nn.py:59-81)."""
    out = copy.deepcopy(next(iter(ensemble.values())))       # (ensemble[0] in the reference; a rank's shard may lack it)
    for name in out:
        if name.split('_')[-1] in ("mean", "var", "tracked"):
            continue
        stack = [copy.deepcopy(sd[name]) for sd in ensemble.values()]
        out[name].copy_(sum(stack) / float(len(stack)))
    return out


def sample_weights(ensemble: Dict[int, Dict[str, torch.Tensor]], n_samples: int = 30
                   ) -> Dict[int, Dict[str, torch.Tensor]]:
    """theta_i ~ N(mu_i, sigma_i) per trainable tensor from the members' mean / std (utils/nn.py:84-117)."""
    out = {i: copy.deepcopy(ensemble[0]) for i in range(n_samples)}
    for name in ensemble[0]:
        if name.split('_')[-1] in ("mean", "var", "tracked"):
            continue
        w_all = torch.cat([copy.deepcopy(m[name])[None, ...] for m in ensemble.values()], dim=0)
        if w_all.dtype == torch.float32:
            ndist = torch.distributions.Normal(torch.mean(w_all, axis=0), torch.std(w_all, axis=0))
            for i in range(n_samples):
                out[i][name].copy_(ndist.sample())
    return out


def gpu_usage_map(cuda_device: int = 0):
    """[used, total] MiB of the device.  The reference shells out to nvidia-smi (utils/nn.py:120-133),
    which does not exist on ROCm; torch's allocator query is used instead."""
    free, total = torch.cuda.mem_get_info(cuda_device)
    return [int((total - free) / 2 ** 20), int(total / 2 ** 20)]
