"""Unit checks (emulator) of kernels whose multi-stage paths only trigger at large sizes in the nets."""
import os
import sys

import numpy as np
import pytest
import torch

from _knobs import set_knob

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module", autouse=True)
def emulator():
    if torch.cuda.is_available():
        pytest.skip("emulator tier is for GPU-less hosts")
    import emu_backend
    emu_backend.use_emulator()


def test_two_stage_bn_statistics_equal_single_stage():
    from atomai_amd import _lib as L
    N, H, W, C = 3, 40, 52, 20
    cs, cop = 20, 32
    x = torch.randn(N, H, W, cs) * 2 + 0.7
    tiles = [(n, ty, tx) for n in range(N) for ty in range(3) for tx in range(4)]
    stats = torch.zeros(len(tiles), 2, cop)
    for i, (n, ty, tx) in enumerate(tiles):
        t = x[n, ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16].reshape(-1, cs).double()
        stats[i, 0, :cs] = t.sum(0).float()
        stats[i, 1, :cs] = ((t - t.mean(0)) ** 2).sum(0).float()
    g, b = torch.rand(C) + 0.5, torch.randn(C)
    outs = []
    for two_stage in (False, True):
        rm, rv = torch.zeros(C), torch.ones(C)
        sc, sh, mu, iv = (torch.empty(cs) for _ in range(4))
        st, rows, mode = stats, len(tiles), 0
        if two_stage:
            nch = 5
            merged = torch.empty(nch, 3, cop)
            L.call("amx_bn_stats_merge", L.ptr(stats), rows, cop, 0, N, H, W, 0, 1, nch, L.ptr(merged), None)
            st, rows, mode = merged, -(-rows // -(-rows // nch)), 2
        L.call("amx_bn_finalize", L.ptr(st), rows, cop, mode, N, H, W, 0, L.ptr(g), L.ptr(b), L.ptr(rm),
               L.ptr(rv), 0.1, 1e-5, C, cs, L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(iv), None)
        outs.append((sc, sh, mu, iv, rm, rv))
    xr = x[..., :C].reshape(-1, C).double()
    np.testing.assert_allclose(outs[0][2][:C].numpy(), xr.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[0][3][:C].numpy(), (1 / (xr.var(0, unbiased=False) + 1e-5).sqrt()).numpy(),
                               rtol=1e-5)
    np.testing.assert_allclose(outs[0][5].numpy(), (0.9 + 0.1 * xr.var(0, unbiased=True)).numpy(), rtol=1e-5)
    for a, c in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a.numpy(), c.numpy(), rtol=2e-6, atol=1e-7)


def test_chunked_row_reduction():
    from atomai_amd import _lib as L
    rows, ncols = 77, 600
    part = torch.randn(rows, ncols)
    nch = 8
    out = torch.empty(nch, ncols)
    L.call("amx_reduce_rows_chunked", L.ptr(part), rows, ncols, nch, L.ptr(out), None)
    chunk = -(-rows // nch)
    used = -(-rows // chunk)
    np.testing.assert_allclose(out[:used].sum(0).numpy(), part.double().sum(0).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout,H,N", [(64, 32, 16, 1), (32, 80, 24, 1), (16, 64, 16, 2)])
def test_conv_layer_wide_channels(cin, cout, H, N):
    """Channel widths of the default nets (wave layouts WM=4/2/1, NT=2, multiple cout blocks) in one layer:
    forward, dgrad, wgrad, BatchNorm backward vs torch autograd in fp64."""
    import copy
    import torch.nn as nn
    from atomai_amd.nets import ConvBlock
    torch.manual_seed(0)
    m = ConvBlock(2, 1, cin, cout, batch_norm=True)
    ref = nn.Sequential(*[copy.deepcopy(l) for l in m.block]).double()
    x = torch.randn(N, cin, H, H)
    x1, x2 = x.clone().requires_grad_(True), x.double().clone().requires_grad_(True)
    y, yr = m(x1), ref(x2)
    gy = torch.randn_like(y)
    y.backward(gy)
    yr.backward(gy.double())
    assert float((y.double() - yr).abs().max()) < 1e-4
    assert float((x1.grad.double() - x2.grad).abs().max() / x2.grad.abs().max()) < 1e-4
    for (k, p), (_, p2) in zip(m.block.named_parameters(), ref.named_parameters()):
        assert float((p.grad.double() - p2.grad).abs().max() / p2.grad.abs().max()) < 1e-4, k


@pytest.mark.parametrize("cin,cout,H", [(16, 32, 40), (32, 16, 36)])
def test_conv_layer_16_row_tiles(cin, cout, H, monkeypatch):
    """The 16-row tile variant forced where the plan would pick 8 rows, with image heights that are not a
    multiple of either tile."""
    set_knob(monkeypatch, "AMX_CONV_TH", "16")
    test_conv_layer_wide_channels(cin, cout, H, 1)


@pytest.mark.parametrize("cin,cout,H", [(64, 32, 20), (16, 16, 36)])
def test_wgrad_4_row_tiles(cin, cout, H, monkeypatch):
    set_knob(monkeypatch, "AMX_WGRAD_TH", "4")
    test_conv_layer_wide_channels(cin, cout, H, 1)


@pytest.mark.parametrize("cin,cout,H,N", [(16, 16, 36, 1), (32, 16, 24, 2), (64, 16, 16, 1), (16, 32, 20, 1),
                                          (16, 64, 16, 1), (32, 32, 24, 1), (32, 80, 16, 1), (64, 64, 20, 1)])
def test_wave_specialised_wgrad_is_bit_identical(cin, cout, H, N, monkeypatch):
    """wgrad_ws.hip (producer / consumer waves, double-buffered LDS images) against wgrad_kernel.h on every wave layout
    of plan_wgrad's plain 3x3 classes — ragged image sides, several tiles per workgroup, bias partials, the fused
    BatchNorm / LeakyReLU backward of the dy loader: the same MFMA order, so the partial rows are bit-identical."""
    import _seg_checks as S
    S.check_wgrad_ws_bit_identical("cpu", cin, cout, H, N, monkeypatch)
    if (cin, cout) == (16, 32):                       # the class whose plan differs: same tile height for both kernels
        S.check_wgrad_ws_bit_identical("cpu", cin, cout, H, N, monkeypatch, force_th=8)
    set_knob(monkeypatch, "AMX_WGRAD_WGS", "3")          # 1-3 workgroups: every one walks several tiles (both LDS images,
    S.check_wgrad_ws_bit_identical("cpu", cin, cout, H, N, monkeypatch)     # the two-tiles-ahead load issue)


def test_adam_flat_kernel_vs_torch_adam_fp64():
    import _adam_checks as A
    A.check_adam_flat_kernel("cpu")
    A.check_adam_flat_kernel("cpu", n=8, steps=3, gscale=1.0)          # no tail
    A.check_adam_flat_kernel("cpu", n=3, steps=3, gscale=0.5)          # tail only


def test_fused_adam_object_vs_torch_adam_fp64():
    import _adam_checks as A
    A.check_fused_adam_vs_torch("cpu")


def test_last_error_names_the_entry_point_and_argument_group():
    from atomai_amd import _lib as L
    t = torch.zeros(16)
    L.load().amx_clear_error()
    assert L.last_error() == ""
    with pytest.raises(L.AmxError) as ei:
        L.call("amx_adam_flat", L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), 0, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.1, 1.0, None)
    assert "amx_adam_flat" in str(ei.value) and "bad argument group" in str(ei.value)
    assert "amx_adam_flat" in L.last_error()


def test_two_tapes_before_one_backward_do_not_alias_the_gradient_bucket():
    """with FusedAdam.prepare() every parameter owns a view of the flat gradient bucket; two module calls
    before ONE backward (loss = f(net(x1)) + f(net(x2))) must still give G1 + G2."""
    import atomai_amd as aoi
    from atomai_amd.optim import FusedAdam
    torch.manual_seed(0)
    blk = aoi.nets.ConvBlock(2, 1, 3, 4, batch_norm=False)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.LeakyReLU(0.01))
    ref[0].load_state_dict({"weight": blk.block[0].weight.detach().clone(), "bias": blk.block[0].bias.detach().clone()})
    x1, x2 = torch.randn(2, 3, 8, 8), torch.randn(2, 3, 8, 8)
    opt = FusedAdam(blk.parameters(), lr=1e-3)
    opt.prepare()
    opt.zero_grad()
    (blk(x1).square().sum() + blk(x2).square().sum()).backward()
    (ref(x1).square().sum() + ref(x2).square().sum()).backward()
    gw, gr = blk.block[0].weight.grad, ref[0].weight.grad
    assert float((gw - gr).abs().max() / gr.abs().max()) < 1e-5
    # and the single-call case still writes straight into the bucket (no copy in step())
    opt.zero_grad()
    blk(x1).square().sum().backward()
    assert blk.block[0].weight.grad.data_ptr() == blk.block[0].weight._amx_grad.data_ptr()


def test_dense_gemm_strides_and_activations():
    import _linear_checks as C
    C.check_gemm_strides("cpu", sizes=((37, 5, 259), (64, 64, 16), (70, 33, 17), (1, 1, 1)))


def test_dense_gemm_split_k():
    import _linear_checks as C
    C.check_gemm_splitk("cpu", sizes=((37, 5, 1100), (40, 24, 1500)))


def test_dense_layer_autograd():
    import _linear_checks as C
    C.check_linear_autograd("cpu")


@pytest.mark.parametrize("bn", [True, False])
def test_convblock_training_dropout(bn):
    import _dropout_checks as D
    D.check_convblock_dropout("cpu", batch_norm=bn)


def test_dilatedblock_dropout_vs_reference_golden():
    import _dropout_checks as D
    D.check_dilated_dropout_golden("cpu")


def test_dilatedblock_without_batchnorm_gradients():
    import _dropout_checks as D
    D.check_dilated_no_batchnorm("cpu")


def test_iou_vs_reference_golden():
    import _metrics_checks as M
    M.check_iou_golden("cpu")


def test_iou_many_classes_and_probabilities_in():
    import _metrics_checks as M
    M.check_iou_wide_golden("cpu")


def test_predict_with_more_than_eight_classes():
    import _seg_checks as C
    C.check_many_classes_predict("cpu")


def test_fit_with_compute_accuracy(tmp_path):
    import _metrics_checks as M
    M.check_fit_with_accuracy(False, tmp_path)



def test_upsample_block_forward_in_one_pass_is_bit_identical():
    import _seg_checks as C
    C.check_upconv_fused_kernel("cpu")

