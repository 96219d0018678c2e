"""`gpu` tier: isolated parity of the fused Adam kernel and optimizer object against torch.optim.Adam (fp64)."""
import pytest

import _adam_checks as A

pytestmark = pytest.mark.gpu


def test_adam_flat_kernel_vs_torch_adam_fp64():
    A.check_adam_flat_kernel("cuda")
    A.check_adam_flat_kernel("cuda", n=594067, steps=10, gscale=0.125)    # the default U-Net's bucket, 1/8 folded in
    A.check_adam_flat_kernel("cuda", n=3, steps=3, gscale=0.5)


def test_fused_adam_object_vs_torch_adam_fp64():
    A.check_fused_adam_vs_torch("cuda")
