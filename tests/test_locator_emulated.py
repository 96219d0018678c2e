"""`not gpu` tier for the Locator kernels (CPU SIMT emulator) vs the reference golden and the scipy oracle."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _locator_checks as K  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


@pytest.mark.parametrize("name", K.GOLDEN_CASES)
def test_reference_golden(name):
    K.check_golden(name, "cpu")


def test_irregular_components():
    K.check_shapes("cpu")


def test_segmentor_predict_with_coordinates():
    K.check_segmentor_predict("cpu")
