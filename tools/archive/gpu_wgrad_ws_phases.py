"""Where do the consumer / producer waves of wgrad_ws.hip spend their time?  (dev tool, round 4)
Uses lib/libatomai_amd_wprof.so (tools/build_variant_lib.sh wprof "-DAMX_WGRAD_PROFILE" wgrad wgrad_ws): every wave
accumulates shader clocks (s_memtime) per phase.  Consumers: sweep / barrier wait; producers: stage (incl. the wait for the
tile's loads) / issue / barrier wait."""
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


def run(N, H, C0, C1, Cout):
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    dpre = torch.randn(N, H, H, Cout, device=dev)
    rows = lib.amx_conv2d_wgrad_rows(N, H, H, C0 + C1, Cout, 9, 1)
    ks = lib.amx_conv2d_wgrad_ksplit(N, H, H, C0 + C1, Cout, 9, 1)
    part = torch.empty(rows, 9, r16(C0 + C1), r16(Cout), device=dev)
    prof = torch.zeros(8192 * 12 * 8, dtype=torch.int64, device=dev)
    lib.amx_wgrad_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))

    def go():
        rc = lib.amx_conv2d_wgrad(L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), None, None, C1, L.ptr(dpre), Cout,
                                  L.ptr(part), N, H, H, Cout, 9, 1, L.stream_ptr(dpre))
        assert rc == 0
    for _ in range(3): go()
    torch.cuda.synchronize()
    prof.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib.amx_wgrad_set_profile_buffer(None)
    t = prof.cpu().numpy().reshape(-1, 12, 8).astype(np.float64)
    t = t[t[:, 0, 7] > 0]
    cons, prod = t[:, :4].reshape(-1, 8), t[:, 4:].reshape(-1, 8)
    prod = prod[prod[:, 7] > 0]                     # 4 or 8 producer waves
    print(f"   ({prod.shape[0] // t.shape[0]} producer waves per workgroup)")
    tiles = cons[:, 6].sum()
    tf = 2.0 * N * H * H * (C0 + C1) * Cout * 9 / ms / 1e9
    print(f"== wgrad_ws {C0}+{C1}->{Cout} @{H} B={N}: {ms*1e3:.1f} us = {tf:.1f} TFLOP/s ({tf/157.3:.3f}); split-K {ks}, "
          f"{t.shape[0]} workgroups, {cons[:, 6].mean():.1f} tiles per wave, lifetime median {np.median(cons[:, 7]):.0f} clocks")
    for who, arr, idx in (("consumer", cons, ((0, "MFMA sweep"), (1, "barrier wait"))),
                          ("producer", prod, ((2, "stage (+ wait for loads)"), (3, "issue loads"), (4, "barrier wait")))):
        for i, name in idx:
            per = arr[:, i].sum() / tiles / (arr.shape[0] / cons.shape[0])
            print(f"   {who} {name:28s} {100 * arr[:, i].sum() / arr[:, 7].sum():5.1f} % of lifetime   {per:9.0f} clocks per tile")


for shape in [(512, 16, 16, 16), (256, 16, 0, 32), (256, 32, 0, 32), (256, 32, 32, 32), (128, 64, 0, 64), (64, 128, 0, 128)]:
    run(32, *shape)
