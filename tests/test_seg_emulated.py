"""`not gpu` tier: the kernel SOURCES (atomai_amd/csrc/*.hip) compiled for the CPU SIMT emulator
(tests/emu) and driven through the real host code (engine, nets, losses, optimizer, trainers), checked
against the reference goldens.  Catches index/layout/logic errors without a GPU; the `gpu` tier repeats
the same checks on the MI355X binary."""
import os
import sys

import numpy as np
import pytest
import torch

from _knobs import set_knob

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import _seg_checks as C  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


@pytest.mark.parametrize("name", list(C.CASES))
def test_net_fwd_bwd_adam(name):
    C.check_net_case(name, "cpu")


@pytest.mark.parametrize("model,ncls,kw", [("SegResNet", 3, dict(batch_norm=False)),
                                           ("SegResNet", 1, dict(layers=[1, 3, 1])),
                                           ("Unet", 3, dict(batch_norm=False))])
def test_variants_vs_oracle(model, ncls, kw):
    C.check_vs_oracle_small(model, ncls, "cpu", **kw)


def test_blocks():
    C.check_blocks("cpu")


@pytest.mark.parametrize("mode", [0, 15])
def test_backward_fusion_modes(mode, monkeypatch):
    """The optional backward fusions (dpre formed in the dgrad / wgrad loaders, BN-backward sums from the dgrad
    epilogue) are off by default (measured slower on MI355X) but stay correct."""
    from atomai_amd import engine
    monkeypatch.setattr(engine, "FUSE", mode)
    C.check_net_case("seg_unet_c3_nf4_b2_32", "cpu")
    C.check_net_case("seg_dilnet_c1_nf5_b2_32", "cpu")


def test_predictor():
    C.check_predict(False)


def test_segmentor_fit_api(tmp_path):
    """API conformance of Segmentor.fit on the emulator: loss goes down, checkpoint format, determinism."""
    import atomai_amd as aoi
    rs = np.random.RandomState(0)
    X = rs.rand(4, 16, 16).astype(np.float32)
    y = rs.randint(0, 3, (4, 16, 16))
    runs = []
    for _ in range(2):
        m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4, upsampling="nearest", seed=1)
        m.fit(X, y, X, y, training_cycles=4, batch_size=2, plot_training_history=False,
              filename=str(tmp_path / "m"))
        runs.append((list(m.loss_acc["train_loss"]), {k: v.clone() for k, v in m.net.state_dict().items()}))
    assert runs[0][0] == runs[1][0]                                  # run-twice determinism
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    assert runs[0][0][-1] < runs[0][0][0]
    assert str(m.criterion) == "CrossEntropyLoss()"
    ck = torch.load(str(tmp_path / "m_metadict_final.tar"), weights_only=False)
    assert ck["model"] == "Unet" and ck["nb_classes"] == 3 and "optimizer" in ck
    assert list(ck["weights"].keys()) == list(m.net.state_dict().keys())
    with pytest.raises(AssertionError):
        aoi.models.Segmentor("Unet", nb_classes=1).fit(X, y, X, y, training_cycles=1, batch_size=2)


def test_activations_are_freed_without_the_garbage_collector():
    """Graph nodes and activations must not form reference cycles: with cycles, every step's activations
    (gigabytes at the benchmark size) stay allocated until Python's cyclic GC happens to run, and the step time
    becomes erratic (the caching allocator has to hipMalloc fresh blocks meanwhile)."""
    import gc
    import weakref
    import atomai_amd as aoi
    from atomai_amd import engine
    rs = np.random.RandomState(0)
    net, _ = aoi.nets.init_fcnn_model("Unet", 3, nb_filters=4)
    x = torch.from_numpy(rs.rand(2, 1, 16, 16).astype(np.float32))
    made, orig = [], engine.Act.__init__

    def tracking_init(self, *a, **k):
        orig(self, *a, **k)
        made.append(weakref.ref(self.t))
    gc.collect()
    gc.disable()
    engine.Act.__init__ = tracking_init
    try:
        net.train()
        y = net(x)
        y.sum().backward()
        del y
        assert len(made) > 10 and sum(r() is not None for r in made) == 0      # incl. the module output
        made.clear()
        net.eval()
        with torch.no_grad():
            y = net(x)
        del y
        assert len(made) > 10 and sum(r() is not None for r in made) == 0
    finally:
        engine.Act.__init__ = orig
        gc.enable()


@pytest.mark.parametrize("opts", [dict(full_epoch=True), dict(swa=True, full_epoch=True, training_cycles=5),
                                  dict(lr_scheduler=[1e-3, 5e-4, 1e-4]),
                                  dict(perturb_weights=True, batch_norm=False),
                                  dict(optimizer=lambda p: torch.optim.SGD(p, lr=1e-2))])
def test_trainer_options(tmp_path, opts):
    """compile_trainer's options (reference: trainers/trainer.py:441-565; test/trainers/test_trainer.py): full-epoch
    data loaders, stochastic weight averaging, learning-rate schedule, time-dependent weight perturbation, and a stock
    torch optimizer driving the HIP modules through ordinary autograd."""
    import warnings
    import atomai_amd as aoi
    opts = dict(opts)
    rs = np.random.RandomState(0)
    X = rs.rand(6, 16, 16).astype(np.float32)
    y = rs.randint(0, 3, (6, 16, 16))
    net_kw = dict(batch_norm=opts.pop("batch_norm")) if "batch_norm" in opts else {}
    m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4, **net_kw)
    init = {k: v.clone() for k, v in m.net.state_dict().items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cycles = opts.pop("training_cycles", 3)
        m.fit(X, y, X, y, training_cycles=cycles, batch_size=2, plot_training_history=False,
              filename=str(tmp_path / "m"), **opts)
    assert len(m.loss_acc["train_loss"]) == cycles and all(np.isfinite(m.loss_acc["train_loss"]))
    moved = [k for k, v in m.net.state_dict().items() if v.is_floating_point() and not torch.equal(v, init[k])]
    assert moved, "training changed nothing"
    if "lr_scheduler" in opts:
        assert m.optimizer.param_groups[0]["lr"] == 1e-4
    if opts.get("swa"):      # the last 5 epochs are averaged (like the reference, SWA needs >= 5 epochs / 30 iterations)
        assert sorted(m.running_weights) == [0, 1, 2, 3, 4]


def test_dilated_layers_on_ragged_sizes():
    C.check_dilated_ragged("cpu")


def test_dilated_layers_halo_class_kernels_still_agree(monkeypatch):
    """AMX_CONV_LATTICE=0 keeps the halo-class kernels of conv_fwd_dil.hip reachable (in-process A/B switch)."""
    set_knob(monkeypatch, "AMX_CONV_LATTICE", "0")
    C.check_dilated_ragged("cpu", cases=((16, 20, 23, 41, 1),))


def test_input_normalisation_inside_the_first_layer_kernel():
    C.check_input_norm_fusion("cpu")


def test_classification_head_in_the_last_conv_epilogue():
    C.check_head_fusion("cpu", wide=False)       # the default widths run on the gpu tier


def test_dilated_block_sum_in_the_last_conv_epilogue():
    C.check_dsum_fusion("cpu")


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 16), (16, 16)])
def test_wave_specialised_thin_conv(cin, cout, monkeypatch):
    """conv_ws.hip against the general kernel and fp64 autograd (32x32 images, 4 emulated CUs)."""
    C.check_wave_specialised_conv("cpu", cin, cout, monkeypatch)


def test_wave_specialised_two_source_layer(monkeypatch):
    C.check_wave_specialised_concat("cpu", monkeypatch)


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 16), (16, 16)])
def test_bn_backward_formed_in_the_loaders(cin, cout, monkeypatch):
    """amx_conv2d_dgrad_fused / amx_conv2d_wgrad_fused against amx_bn_bwd_apply + the plain kernels (bit-identical)."""
    C.check_bwd_fused_in_loaders("cpu", cin, cout, monkeypatch)


def test_bn_backward_formed_in_the_loaders_unet(monkeypatch):
    """U-Net nb_filters 16 at 64x64: c6.0 (two outputs), c5.3 / c2.3 (32 -> 32) and c2.0 (32 -> 16) take the fused path."""
    assert C.check_bwd_fused_in_loaders("cpu", 0, 0, monkeypatch, hw=64, batch=2, unet=True) == 4



@pytest.mark.parametrize("cin,cout", [(8, 16), (32, 32)])
def test_bn_backward_formed_in_the_loaders_resblock(cin, cout, monkeypatch):
    """ResBlocks through the fused BatchNorm-backward loaders (c1 and c2 of every block) against the two-pass form."""
    assert C.check_bwd_fused_in_loaders("cpu", cin, cout, monkeypatch, hw=32, batch=2, res=True, repeats=2) >= 2


def test_loss_upstream_gradient_factor():
    C.check_loss_upstream_gradient("cpu")


def test_eval_pool_inside_the_first_layer_kernel():
    C.check_pool_fusion("cpu")


def test_upsample_forward_is_exact():
    C.check_upsample_exact("cpu")


def test_remainder_column_classes_vs_padded_plan():
    """25 / 50-filter layers on the 16 + 3 x 4 / 3 x 16 + 4 column plan (v_mfma_f32_4x4x1 remainder blocks)."""
    C.check_remainder_columns("cpu", cases=((50, 50, 1, 20, 1), (25, 50, 2, 18, 1), (50, 25, 1, 16, 1), (28, 50, 6, 21, 1)))


def test_upsample_block_as_one_launch_inside_the_nets():
    C.check_upconv_node("cpu")


def test_hooked_block_by_block_forward_equals_fused():
    C.check_hooked_forward_equals_fused("cpu")


def test_lattice_xpack_is_bit_identical(monkeypatch):
    C.check_lattice_xpack_bit_identical("cpu", monkeypatch, cases=((52, 50, 20, 70, 1, 6), (28, 50, 13, 29, 1, 4)))


def test_two_source_data_gradient_as_two_wave_specialised_launches(monkeypatch):
    C.check_split_two_source_dgrad("cpu", monkeypatch)


def test_head_and_loss_of_the_training_step_in_one_pass():
    C.check_fused_head_and_loss("cpu")


def test_pool_backward_fused_with_the_first_layer_weight_gradient():
    C.check_pool_backward_with_first_layer_wgrad("cpu")
