"""`gpu` tier: checkpoint interchange with the reference through libatomai_amd.so on the MI355X."""
import pytest

import _ckpt_checks as C

pytestmark = pytest.mark.gpu


def test_load_reference_segmentor_checkpoint():
    C.check_load_reference_seg()


def test_load_reference_rvae_checkpoint():
    C.check_load_reference_rvae()


@pytest.mark.parametrize("model", ["Unet", "dilnet", "SegResNet", "ResHedNet"])
def test_io_segmentor(tmp_path, model):
    C.check_roundtrip_seg(tmp_path, model)


def test_io_rvae(tmp_path):
    C.check_roundtrip_rvae(tmp_path)


def test_ensemble_trainer_matches_reference(tmp_path):
    import _ensemble_checks as E
    E.check_ensemble(tmp_path, tol=3e-2)


def test_ensemble_predictor_matches_reference():
    import _ensemble_checks as E
    E.check_ensemble_predictor()
