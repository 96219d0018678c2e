"""`not gpu` tier: checkpoint interchange with the reference (kernels run on the CPU SIMT emulator)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _ckpt_checks as C  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


def test_load_reference_segmentor_checkpoint():
    C.check_load_reference_seg()


def test_load_reference_rvae_checkpoint():
    C.check_load_reference_rvae()


@pytest.mark.parametrize("model", ["Unet", "dilnet", "SegResNet", "ResHedNet"])
def test_io_segmentor(tmp_path, model):
    C.check_roundtrip_seg(tmp_path, model)


def test_io_rvae(tmp_path):
    C.check_roundtrip_rvae(tmp_path)


def test_misc_loaders(tmp_path):
    C.check_misc_loaders(tmp_path)


def test_ensemble_trainer_matches_reference(tmp_path):
    import _ensemble_checks as E
    E.check_ensemble(tmp_path, tol=5e-4)


def test_ensemble_predictor_matches_reference():
    import _ensemble_checks as E
    E.check_ensemble_predictor()
