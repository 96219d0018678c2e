"""`not gpu` tier for the DKL covariance path: HIP kernel sources on the SIMT emulator vs the float64 oracle
(oracle/gp_oracle.py — closed forms, PARITY UNPINNED w.r.t. gpytorch, see its header) and vs torch autograd
of the same closed forms."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _gp_checks as G  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", ["rbf", "matern"])
def test_kernel_matrix_and_matvec(dtype, kind):
    G.check_kernel_matrix("cpu", dtype, kind, N=70, M=45, D=3)
    G.check_kernel_matrix("cpu", dtype, kind, N=33, M=260, D=2)


@pytest.mark.parametrize("kind", ["rbf", "matern"])
def test_kernel_backward_and_mll(kind):
    G.check_mll_and_grads("cpu", kind, N=50, D=2)


def test_dklgpr_api():
    G.check_dklgpr_api()


def test_gpytorch_known_answer_vectors():
    G.check_gpytorch_known_answers_oracle()
    G.check_gpytorch_known_answers_kernel("cpu")
    G.check_scale_to_bounds_module("cpu")


def test_posterior_is_factorised_once_per_model_state():
    G.check_posterior_cache("cpu")


def test_conv_feature_extractor_vs_stock_torch():
    G.check_conv_feature_extractor("cpu")


def test_dklgpr_with_conv_feature_extractor():
    G.check_dklgpr_conv_extractor("cpu", N=48, p=8, cycles=2, precision="double")
