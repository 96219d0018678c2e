"""bench.py's algorithmic FLOP count (the numerator of every roofline fraction it prints) is pinned to SURVEY.md
section 8d's figures and to the convolutions the engine really launches (counted on the CPU emulator)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_step_flops_match_survey():
    b = _bench()
    # 1x1 up-convs evaluated at low resolution -> 16.609 GFLOP/img fwd, 49.75 GFLOP/img fwd+bwd,
    # 1.592 TFLOP per bs-32 step
    assert abs(b.step_flops(32) / 1e12 - 1.592) < 1e-3
    assert abs(b.step_flops(1) / 1e9 - 49.75) < 0.01


def test_launched_conv_flops_match_the_table():
    """Count 2*Cin*Cout*taps*N*H*W over the conv entry points of ONE training step of the default U-Net at a small
    size (emulator) and compare with the table scaled to that size: nothing is skipped, nothing is counted twice."""
    if torch.cuda.is_available():
        pytest.skip("emulator tier")
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    from atomai_amd import _lib as L
    import atomai_amd.engine as eng
    b = _bench()
    B, H = 1, 16
    net, _ = aoi.nets.init_fcnn_model("Unet", 3)
    x = torch.from_numpy(np.random.RandomState(0).rand(B, 1, H, H).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(1).randint(0, 3, (B, H, H)))
    flops, orig = [0.0], L.call

    def call(name, *a):
        if name == "amx_conv2d_fwd":
            flops[0] += 2.0 * (a[3] + a[7]) * a[19] * a[20] * a[16] * a[17] * a[18]
        elif name == "amx_conv2d_dgrad":
            flops[0] += 2.0 * a[1] * (a[5] + a[7]) * a[11] * a[8] * a[9] * a[10]
        elif name == "amx_conv2d_dgrad_fused":        # (the wave-specialised data gradient of large thin layers)
            flops[0] += 2.0 * a[6] * (a[9] + a[11]) * a[15] * a[12] * a[13] * a[14]
        elif name == "amx_conv2d_dgrad_fused_bsum":   # (the same + BatchNorm-backward sums of the source layer, round 6)
            flops[0] += 2.0 * a[6] * a[9] * a[13] * a[10] * a[11] * a[12]
        elif name == "amx_conv2d_wgrad_fused":
            flops[0] += 2.0 * (a[3] + a[7]) * a[20] * a[21] * a[17] * a[18] * a[19]
        return orig(name, *a)
    L.call = eng.L.call = call
    try:
        net.train()
        loss = aoi.losses_metrics.select_loss("ce", 3)(net(x), y)
        loss.backward()
    finally:
        L.call = eng.L.call = orig
    # bench.step_flops is the algorithmic count of the WHOLE step (SURVEY's figure); the MFMA entry points cover all
    # of it except the two VALU layers: c1 (Cin = 1: forward + weight gradient) and the px head (forward, data and
    # weight gradient)
    valu = B * H * H * (2 * (2.0 * 1 * 16 * 9) + 3 * (2.0 * 16 * 3 * 1))
    expected = b.step_flops(B) * (H * H) / (512 * 512) - valu
    assert abs(flops[0] - expected) / expected < 1e-6, (flops[0], expected)
