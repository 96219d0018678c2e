"""Phase clocks of the wave-specialised thin-layer kernel (conv_ws.hip) on the MI355X.  Needs lib/libatomai_amd_prof.so:
   tools/build_variant_lib.sh prof "-DAMX_CONV_PROFILE" conv_fwd conv_ws"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L

LIBNAME = sys.argv[1] if len(sys.argv) > 1 else "prof"
print("library variant:", LIBNAME)
lib = ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{LIBNAME}.so"))
for name in ("amx_conv2d_fwd", "amx_pack_weights", "amx_pack_weights_size", "amx_conv2d_num_tiles", "amx_conv2d_tile_h"):
    fn = getattr(lib, name); fn.restype, fn.argtypes = L.SIGNATURES[name]
lib.amx_conv_set_profile_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
NAMES = ["MFMA sweep", "wait at barrier A", "hand-over (bias, act, ds_write)", "wait at barrier B", "staging (wait loads, affine, ds_write)",
         "issue loads", "epilogue (statistics, stores)", "lifetime"]


def run(tag, N, H, C0, C1, Cout, Y1=0, stats_on=True):
    torch.manual_seed(0)
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) / ((C0 + C1) * 9) ** 0.5
    X0 = torch.randn(N, H, H, C0, device=dev)
    X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    wpk = torch.empty(lib.amx_pack_weights_size(Cout, C0, C1, 9, 0), device=dev)
    lib.amx_pack_weights(L.ptr(w), L.ptr(wpk), Cout, C0, C0, C1, C1, 9, 0, L.stream_ptr(w))
    bias = torch.randn(Cout, device=dev) if stats_on else None
    Y0 = Cout - Y1
    y = torch.zeros(N, H, H, Y0, device=dev)
    y1 = torch.zeros(N, H, H, Y1, device=dev) if Y1 else None
    stats = None
    if stats_on:
        th = lib.amx_conv2d_tile_h(C0 + C1, Cout, 9, 1, H)
        stats = torch.zeros(lib.amx_conv2d_num_tiles(N, H, H, th), 2, Cout, device=dev)
    prof = torch.zeros(256 * 16 * 8, dtype=torch.int64, device=dev)

    def go():
        rc = lib.amx_conv2d_fwd(L.ptr(X0), L.ptr(sc) if stats_on else None, L.ptr(sh) if stats_on else None, C0, L.ptr(X1), None, None, C1,
                                L.ptr(wpk), L.ptr(bias), None, L.ptr(y), Y0, L.ptr(y1), Y1, L.ptr(stats), N, H, H, Cout, 9, 1,
                                0.01 if stats_on else 1.0, L.stream_ptr(y))
        assert rc == 0
    for _ in range(3): go()
    lib.amx_conv_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib.amx_conv_set_profile_buffer(None)
    t = prof.cpu().numpy().reshape(256, 16, 8).astype(np.float64)
    tiles = (H // 16) ** 2 * N / 256.0
    print(f"== {tag}: {ms * 1e3:.1f} us, {tiles:.0f} tiles per workgroup; clocks per tile (median over waves), s_memtime ticks")
    for role, sl, idx in (("consumer", slice(0, 8), (0, 1, 2, 3, 7)), ("producer", slice(8, 16), (4, 5, 6, 1, 3, 7))):
        life = np.median(t[:, sl, 7])
        for i in idx:
            v = np.median(t[:, sl, i])
            print(f"   {role}  {NAMES[i]:42s} {v / tiles:9.0f}   {100 * v / life:5.1f} %")


run("c2a fwd 16->32 @256", 32, 256, 16, 0, 32)
run("c2b fwd 32->32 @256", 32, 256, 32, 0, 32)
run("c6 fwd 16+16->16 @512", 32, 512, 16, 16, 16)
run("c6 dgrad 16->16|16 @512", 32, 512, 16, 0, 32, Y1=16, stats_on=False)
