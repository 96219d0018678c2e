"""`gpu` tier for the on-device augmentation."""
import pytest

import _aug_checks as A

pytestmark = pytest.mark.gpu


def test_kernels_match_reference_golden():
    A.check_kernels_vs_reference_golden("cuda")


def test_kernels_match_oracle_on_every_step():
    A.check_kernels_vs_oracle_all_steps("cuda")
    A.check_kernels_vs_oracle_all_steps("cuda", N=7, H=96, W=96)


def test_generator_statistics():
    A.check_generator_statistics("cuda", N=8, H=256, W=256)


def test_class_drop_and_trainer_hook():
    A.check_class_drop_and_trainer_hook("cuda")


def test_zoom_and_resize_vs_reference_golden():
    A.check_augment_geometry_golden("cuda")


def test_img_resize_and_predictor_resize():
    A.check_img_resize("cuda")


def test_custom_transform_host_callable():
    A.check_custom_transform("cuda")
