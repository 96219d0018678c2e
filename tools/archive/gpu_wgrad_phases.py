"""Where does a wave of the weight-gradient kernel spend its time?  (dev tool)
Uses lib/libatomai_amd_wprof.so (tools/build_variant_lib.sh wprof "-DAMX_WGRAD_PROFILE" wgrad): every wave accumulates
shader clocks (s_memtime) per phase of its tile loop.  Prints, per layer shape of the bs-32 U-Net step, the share of a
wave's lifetime in each phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L

lib = ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_wprof.so"))
for name in ("amx_conv2d_wgrad", "amx_conv2d_wgrad_rows", "amx_conv2d_wgrad_ksplit"):
    fn = getattr(lib, name); fn.restype, fn.argtypes = L.SIGNATURES[name]
lib.amx_wgrad_set_profile_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
r16 = lambda v: (v + 15) // 16 * 16
PH = ["stage (affine + LDS writes)", "barrier after stage", "issue next tile's loads", "MFMA sweep", "barrier after MFMA", "epilogue (partial rows)"]


def run(N, H, C0, C1, Cout, taps):
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    dpre = torch.randn(N, H, H, Cout, device=dev)
    rows = lib.amx_conv2d_wgrad_rows(N, H, H, C0 + C1, Cout, taps, 1)
    ks = lib.amx_conv2d_wgrad_ksplit(N, H, H, C0 + C1, Cout, taps, 1)
    part = torch.empty(rows, taps, r16(C0 + C1), r16(Cout), device=dev)
    prof = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device=dev)
    lib.amx_wgrad_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))

    def go():
        rc = lib.amx_conv2d_wgrad(L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), None, None, C1, L.ptr(dpre), Cout,
                                  L.ptr(part), N, H, H, Cout, taps, 1, L.stream_ptr(dpre))
        assert rc == 0
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib.amx_wgrad_set_profile_buffer(None)
    t = prof.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t[:, 7] > 0]
    life, tiles = t[:, 7], t[:, 6]
    tf = 2.0 * N * H * H * (C0 + C1) * Cout * taps / ms / 1e9
    print(f"== wgrad {C0}+{C1}->{Cout} @{H} taps {taps} B={N}: {ms*1e3:.1f} us = {tf:.1f} TFLOP/s ({tf/157.3:.3f}); split-K {ks}, "
          f"{t.shape[0]} waves, {tiles.mean():.1f} tiles per wave, wave lifetime median {np.median(life):.0f} clocks")
    for i, name in enumerate(PH):
        per_tile = t[:, i].sum() / max(tiles.sum(), 1) if i < 5 else t[:, i].mean()
        print(f"   {name:32s} {100 * t[:, i].sum() / life.sum():5.1f} % of lifetime   {per_tile:9.0f} clocks per {'tile' if i < 5 else 'wave'}")


for shape in [(512, 16, 16, 16, 9), (256, 32, 0, 32, 9), (256, 32, 32, 32, 9), (128, 64, 0, 64, 9), (128, 64, 64, 64, 9), (64, 128, 0, 128, 9)]:
    run(32, *shape)
