"""`not gpu` tier for the VAE / rVAE path: kernel sources on the CPU SIMT emulator vs the reference goldens.

Not executed here (device-only instruction paths; the `gpu` tier runs the same checks on them): the hardware
exp / reciprocal INSTRUCTIONS of tanh in csrc/rdecoder.hip (`rd_tanh`; the emulator build runs the library tanhf by
default and, under amx_emu_set_fast_tanh(1), the same algebraic form 1 - 2 / (e^2x + 1) on expf / division —
`test_fast_tanh_form_*` below) and the `v_exp_f32` form of the RBF kernel in csrc/kernel_matrix.hip (emulator: expf).
Index / layout / reduction logic is identical in both builds."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _vae_checks as V  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


@pytest.mark.parametrize("name", list(V.CASES))
def test_elbo_grads_adam(name):
    V.check_vae_case(name, "cpu")


@pytest.mark.parametrize("hid,nl,skip,hw", [(64, 1, 0, (7, 5)), (128, 2, 0, (12, 12)), (128, 3, 1, (8, 8)),
                                            (32, 5, 0, (9, 6, 2)), (100, 4, 1, (6, 6, 3)),
                                            # beyond the fused kernels (width > 128, > 5 layers, > 4 channels): layer by layer
                                            (160, 2, 0, (7, 5)), (24, 6, 1, (6, 5)), (32, 2, 0, (5, 4, 5))])
def test_rdecoder_shapes(hid, nl, skip, hw):
    V.check_rdecoder_shapes("cpu", hid, nl, skip, hw)


def test_rvae_fit_api(tmp_path):
    import atomai_amd as aoi
    X = np.random.RandomState(0).rand(8, 8, 8).astype(np.float32)
    m = aoi.models.rVAE((8, 8), latent_dim=2, numhidden_encoder=32, numhidden_decoder=32)
    m.fit(X, training_cycles=2, batch_size=4, filename=str(tmp_path / "rv"))
    assert len(m.loss_history["train_loss"]) == 2
    ck = torch.load(str(tmp_path / "rv.tar"), weights_only=False)
    assert {"encoder", "decoder", "optimizer", "num_iter"} <= set(ck.keys())
    assert m.decode(np.array([0.0, 0.0], dtype=np.float32)).shape == (1, 8, 8)


def test_subimage_utilities_and_encode_images(golden_dir):
    """Bridge utilities between Segmentor output and VAE input (utils/img.py: extract_subimages, get_coord_grid,
    crop_borders) and BaseVAE.encode_image_ / encode_images / reconstruct, against the reference golden."""
    from collections import OrderedDict
    import atomai_amd as aoi
    from atomai_amd.utils import crop_borders, extract_subimages, get_coord_grid
    g = np.load(os.path.join(golden_dir, "vae_api.npz"))
    coords = {k: g[f"coords|{k}"] for k in (0, 1)}
    for ws in (5, 6):
        st, com, fr = extract_subimages(g["img"], coords, ws)
        assert np.array_equal(st, g[f"sub|{ws}|stack"]) and np.array_equal(com, g[f"sub|{ws}|com"])
        assert np.array_equal(fr, g[f"sub|{ws}|frames"])
    assert np.array_equal(get_coord_grid(g["img"][..., 0], 3, return_dict=False), g["grid3"])
    assert np.array_equal(get_coord_grid(g["img"][0, ..., 0], 7)[0], g["grid_dict2"])
    assert np.array_equal(crop_borders(g["crop_in"], -1), g["crop_out"])
    st, com, fr = extract_subimages(g["img"], {0: np.zeros((0, 3)), 1: np.array([[0.0, 0.0, 0.0]])}, 5)
    assert len(st) == 0                                            # nothing fits: empty lists, as the reference

    v = aoi.models.VAE((8, 8), latent_dim=2, seed=0, numhidden_encoder=16, numhidden_decoder=16)
    v.encoder_net.load_state_dict(OrderedDict((k[4:], torch.from_numpy(g[k])) for k in g.files if k.startswith("enc|")))
    v.decoder_net.load_state_dict(OrderedDict((k[4:], torch.from_numpy(g[k])) for k in g.files if k.startswith("dec|")))
    im_, enc_ = v.encode_image_(g["big"], num_batches=3)
    assert np.array_equal(im_, g["encimg|img"])
    np.testing.assert_allclose(enc_, g["encimg|z"], rtol=1e-4, atol=1e-6)
    ims, encs = v.encode_images(np.stack([g["big"], g["big"][::-1].copy()]), num_batches=4)
    assert np.array_equal(ims, g["encimgs|img"])
    np.testing.assert_allclose(encs, g["encimgs|z"], rtol=1e-4, atol=1e-6)
    torch.manual_seed(3)
    rec = v.reconstruct(g["big"][:8, :8][None], num_samples=4)
    assert rec.shape == g["recon"].shape


@pytest.mark.parametrize("hid,nl,skip,hw", [(32, 2, 0, (9, 7)), (64, 3, 1, (12, 11, 2))])
def test_rdecoder_saved_activations_equal_recompute(hid, nl, skip, hw):
    V.check_rdecoder_saved_equals_recompute("cpu", hid, nl, skip, hw)


def test_rvae_fused_latent_and_scalar_elbo_path():
    V.check_rvae_fused_latent_path("cpu")


@pytest.fixture
def fast_tanh():
    import emu_backend
    lib = emu_backend.use_emulator()
    lib.amx_emu_set_fast_tanh(1)
    yield
    lib.amx_emu_set_fast_tanh(0)


def test_fast_tanh_form_elementwise_bound(fast_tanh):
    """The cancellation structure of the device's rd_tanh (1 - 2 / (e^2x + 1)) on the CPU tier: absolute error <= 2e-7
    with a correctly rounded exp (the device's v_exp_f32 adds ~1 ulp of e^2x: the gpu tier holds it to the same bound)."""
    V.check_rd_tanh_bound("cpu", 2e-7)


@pytest.mark.parametrize("name", ["rvae16", "rvae12_rgb"])
def test_fast_tanh_form_goldens(name, fast_tanh):
    V.check_vae_case(name, "cpu")
