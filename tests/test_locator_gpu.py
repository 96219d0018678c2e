"""`gpu` tier for the Locator kernels through libatomai_amd.so on the MI355X."""
import pytest

import _locator_checks as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", K.GOLDEN_CASES)
def test_reference_golden(name):
    K.check_golden(name, "cuda")


def test_irregular_components_up_to_1024():
    K.check_shapes("cuda", big=True)


def test_segmentor_predict_with_coordinates():
    K.check_segmentor_predict("cuda")
