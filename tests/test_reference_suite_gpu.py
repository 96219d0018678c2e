"""`gpu` tier: the reference's own hot-path test scenarios (tests/_reference_suite.py) through libatomai_amd.so on the MI355X."""
import os
import sys

import pytest
import torch

import _reference_suite as R  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
NARROW = {}                          # default widths on the MI355X


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "gpu tier needs an MI355X"
    from atomai_amd import _lib
    _lib.load()
    assert not _lib.is_test_backend()


def test_trainer_loss_selection():
    R.loss_selection()


@pytest.mark.parametrize("model_type", ["Unet", "dilnet"])
def test_segtrainer_determinism(model_type, tmp_path):
    R.segtrainer_determinism(model_type, tmp_path)


@pytest.mark.parametrize("binary", [True, False])
def test_segtrainer_dataloader(binary):
    R.segtrainer_dataloader(binary, DEV)


@pytest.mark.parametrize("bn,n_bn,layers", [(True, 16, [1, 2, 3, 4]), (True, 20, [2, 3, 3, 4]), (False, 0, [2, 3, 3, 4])])
def test_init_unet_bn(bn, n_bn, layers):
    R.init_unet_bn(bn, n_bn, layers)


@pytest.mark.parametrize("model,dropout,n", [("Unet", False, 0), ("Unet", True, 3), ("dilnet", False, 0), ("dilnet", True, 2)])
def test_init_dropouts(model, dropout, n):
    R.init_dropouts(model, dropout, n)


@pytest.mark.parametrize("layers", [[1, 2, 3, 4], [2, 3, 3, 4]])
def test_init_unet_layers(layers):
    R.init_unet_layers(layers)


@pytest.mark.parametrize("bn,n_bn,layers", [(True, 6, [1, 2, 2, 1]), (True, 10, [2, 3, 3, 2]), (False, 0, [3, 4, 4, 1])])
def test_init_dilnet_bn(bn, n_bn, layers):
    R.init_dilnet_bn(bn, n_bn, layers)


@pytest.mark.parametrize("layers", [[1, 2, 2, 1], [2, 3, 3, 2], [3, 4, 4, 1]])
def test_init_dilnet_layers(layers):
    R.init_dilnet_layers(layers)


@pytest.mark.parametrize("model,nf,expected", [("Unet", 25, [25, 50, 100, 200, 100, 50, 25]), ("dilnet", 25, [25, 50, 50, 25])])
def test_init_segmodel_filters(model, nf, expected):
    R.init_segmodel_filters(model, nf, expected)


@pytest.mark.parametrize("enc", ["fcEncoderNet", "convEncoderNet"])
@pytest.mark.parametrize("dec", ["fcDecoderNet", "convDecoderNet"])
@pytest.mark.parametrize("separately", [False, True])
def test_vi_set_nets(enc, dec, separately):
    R.vi_set_nets(enc, dec, separately)


@pytest.mark.parametrize("torch_format", [True, False])
def test_vi_set_data(torch_format):
    R.vi_set_data(torch_format)


def test_vi_reparametrize():
    R.vi_reparametrize(DEV)


def test_vi_custom_optimizer():
    R.vi_custom_optimizer()


@pytest.mark.parametrize("kind,latent,translation", [("VAE", 2, True), ("VAE", 10, True), ("rVAE", 2, True),
                                                     ("rVAE", 2, False), ("rVAE", 10, True)])
def test_vae_encoding(kind, latent, translation):
    R.vae_encoding(kind, latent, translation)


@pytest.mark.parametrize("kind,conv_e,conv_d,latent,translation,ncls",
                         [("VAE", False, False, 2, True, 0), ("VAE", True, True, 10, True, 0), ("VAE", True, False, 2, True, 3),
                          ("rVAE", False, False, 2, True, 0), ("rVAE", True, False, 10, False, 0), ("rVAE", False, False, 2, True, 3)])
def test_vae_decoding(kind, conv_e, conv_d, latent, translation, ncls):
    R.vae_decoding(kind, conv_e, conv_d, latent, translation, ncls)


@pytest.mark.parametrize("conv_e,conv_d,latent", [(False, False, 2), (True, True, 10)])
def test_vae_reconstruct(conv_e, conv_d, latent):
    R.vae_reconstruct(conv_e, conv_d, latent)


@pytest.mark.parametrize("kind,latent", [("VAE", 2), ("rVAE", 2)])
def test_vae_encode_image(kind, latent):
    R.vae_encode_image(kind, latent)


def test_basepredictor():
    R.basepredictor(DEV)


@pytest.mark.parametrize("model", ["Unet", "dilnet", "SegResNet", "ResHedNet"])
@pytest.mark.parametrize("shape", [(2, 8, 8), (8, 8)])
def test_segpredictor(model, shape):
    R.segpredictor(model, shape, DEV)


@pytest.mark.parametrize("kw", [dict(gauss_noise=True), dict(poisson_noise=[30, 45]), dict(salt_and_pepper=True),
                                dict(blur=True), dict(contrast=True), dict(background=True), dict(jitter=[0, 20]),
                                dict(rotation=True), dict(zoom=True), dict(resize=True),
                                dict(rotation=True, zoom=True, gauss_noise=True, blur=True, contrast=True)])
def test_imaug_transforms(kw):
    R.imaug_individual(kw, DEV)


def test_dkl_fit():
    R.dkl_fit()


@pytest.mark.parametrize("shared_emb", [0, 1])
def test_dkl_fit_ensemble(shared_emb):
    R.dkl_fit_ensemble(shared_emb)


def test_dkl_predict():
    R.dkl_predict()


def test_dkl_multi_model_predict():
    R.dkl_multi_model_predict()


@pytest.mark.parametrize("shared_emb", [0, 1])
@pytest.mark.parametrize("ydim", [(50,), (1, 50)])
def test_dkl_ensemble_predict(shared_emb, ydim):
    R.dkl_ensemble_predict(shared_emb, ydim)


@pytest.mark.parametrize("reg_dim,shared", [(1, True), (2, True), (2, False)])
def test_dkl_sampling_and_thompson(reg_dim, shared):
    R.dkl_sampling(reg_dim, shared)


@pytest.mark.parametrize("precision,dtype", [("single", torch.float32), ("double", torch.float64)])
def test_dkltrainer_precision(precision, dtype):
    R.dkltrainer_precision(precision, dtype)


def test_dkltrainer_compile_train_run(tmp_path):
    R.dkltrainer_compile_train_run(tmp_path)


def test_dkltrainer_multi_model():
    R.dkltrainer_multi_model()


@pytest.mark.parametrize("model", ["Unet", "dilnet", "SegResNet", "ResHedNet"])
def test_io_segmentor(model, tmp_path):
    R.io_segmentor(model, tmp_path, **NARROW)


@pytest.mark.parametrize("kind", ["VAE", "rVAE"])
def test_io_vae_and_resume(kind, tmp_path):
    R.io_vae(kind, tmp_path)


@pytest.mark.parametrize("full_epoch", [0, 1])
@pytest.mark.parametrize("binary", [1, 0])
@pytest.mark.parametrize("model", ["Unet", "dilnet", "SegResNet", "ResHedNet"])
def test_ensemble_seg(model, binary, full_epoch, tmp_path):
    R.ensemble_seg(model, binary, full_epoch, tmp_path, **NARROW)


@pytest.mark.parametrize("model", ["Unet", "dilnet", "ResHedNet"])
def test_epredictor_seg(model, tmp_path):
    R.epredictor_seg(model, tmp_path, **NARROW)
