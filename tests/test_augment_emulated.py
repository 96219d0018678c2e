"""`not gpu` tier for the on-device augmentation (csrc/aug.hip on the SIMT emulator)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _aug_checks as A  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


def test_oracle_matches_reference_golden():
    A.check_oracle_vs_reference_golden()


def test_kernels_match_reference_golden():
    A.check_kernels_vs_reference_golden("cpu")


def test_kernels_match_oracle_on_every_step():
    A.check_kernels_vs_oracle_all_steps("cpu")


def test_generator_statistics():
    A.check_generator_statistics("cpu", N=2, H=48, W=48)


def test_class_drop_and_trainer_hook():
    A.check_class_drop_and_trainer_hook("cpu")


def test_zoom_and_resize_vs_reference_golden():
    A.check_augment_geometry_golden("cpu")


def test_img_resize_and_predictor_resize():
    A.check_img_resize("cpu")


def test_custom_transform_host_callable():
    A.check_custom_transform("cpu")
