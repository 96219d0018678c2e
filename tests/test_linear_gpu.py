"""`gpu` tier: the dense-layer GEMM (csrc/linear.hip) and its autograd wrapper."""
import pytest

import _linear_checks as C

pytestmark = pytest.mark.gpu


def test_dense_gemm_strides_and_activations():
    C.check_gemm_strides("cuda")
    C.check_gemm_strides("cuda", sizes=((512, 128, 4096), (16384, 1000, 256), (16384, 2, 50)))   # rVAE encoder / DKL extractor


def test_dense_layer_autograd():
    C.check_linear_autograd("cuda")


def test_dense_gemm_split_k():
    C.check_gemm_splitk("cuda")
