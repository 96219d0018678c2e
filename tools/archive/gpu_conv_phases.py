"""Where does a wave of the convolution kernel spend its time?  (dev tool)
Uses lib/libatomai_amd_prof.so (tools/build_variant_lib.sh prof "-DAMX_CONV_PROFILE" conv_fwd conv_fwd_3x3 conv_fwd_dil
conv_fwd_1x1): every wave stamps the shader clock (s_memtime) at its phase boundaries.  Prints, per layer shape, the
median duration of each phase over the steady-state workgroups and the share of the wave's lifetime."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L

lib = ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_prof.so"))
for name in ("amx_conv2d_fwd", "amx_pack_weights", "amx_pack_weights_size", "amx_conv2d_num_tiles", "amx_conv2d_tile_h"):
    fn = getattr(lib, name); fn.restype, fn.argtypes = L.SIGNATURES[name]
lib.amx_conv_set_profile_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
PH = ["entry -> kernel args + tile coords", "x_off index math", "issue loads c0 (10 VMEM)", "wait loads + LDS stage c0", "barrier", "issue loads c1", "MFMA c0", "barrier",
      "LDS stage c1", "barrier", "issue (none)", "MFMA c1", "barrier", "bias + activation (registers)", "statistics + LDS transpose + stores"]
SL = [(0, 14), (14, 15), (15, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11), (11, 12), (12, 13)]


def r4(c): return (c + 3) // 4 * 4
def r16(c): return (c + 15) // 16 * 16


def run(N, H, C0, Cout, stats_on=True):
    torch.manual_seed(0)
    C0s, Cos = r4(C0), r4(Cout)
    w = torch.randn(Cout, C0, 3, 3, device=dev) / (C0 * 9) ** 0.5
    X0 = torch.randn(N, H, H, C0s, device=dev)
    sc = torch.rand(C0s, device=dev) + 0.5; sh = torch.randn(C0s, device=dev)
    wpk = torch.empty(lib.amx_pack_weights_size(Cout, C0s, 0, 9, 0), device=dev)
    lib.amx_pack_weights(L.ptr(w), L.ptr(wpk), Cout, C0, C0s, 0, 0, 9, 0, L.stream_ptr(w))
    bias = torch.randn(r16(Cout), device=dev)
    y = torch.empty(N, H, H, Cos, device=dev)
    th = lib.amx_conv2d_tile_h(C0s, Cout, 9, 1, H)              # rows per statistics row = rows per WAVE
    tiles = lib.amx_conv2d_num_tiles(N, H, H, 4 * th)           # workgroup tiles: 4 waves
    nob = -(-r16(Cout) // (16 * (1 if Cout <= 16 else (4 if (Cout >= 64 and C0s >= 128) else 2))))
    stats = torch.zeros(lib.amx_conv2d_num_tiles(N, H, H, th), 2, r16(Cout), device=dev) if stats_on else None
    prof = torch.zeros(tiles * nob * 4 * 16, dtype=torch.int64, device=dev)
    lib.amx_conv_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))

    def go():
        rc = lib.amx_conv2d_fwd(L.ptr(X0), L.ptr(sc), L.ptr(sh), C0s, None, None, None, 0, L.ptr(wpk), L.ptr(bias), None,
                                L.ptr(y), Cos, None, 0, L.ptr(stats), N, H, H, Cout, 9, 1, 0.01, L.stream_ptr(y))
        assert rc == 0
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib.amx_conv_set_profile_buffer(None)
    t = prof.cpu().numpy().reshape(-1, 4, 16).astype(np.float64)
    t = t - t[t > 0].min()
    nwg = t.shape[0]
    t = t[nwg // 4: 3 * nwg // 4]                               # steady state
    nchunk = -(-C0s // 16)
    life = t[:, :, 13 if stats_on else 12] - t[:, :, 0]
    span = (t[..., 12].max() - t[..., 0].min())
    print(f"== {C0}->{Cout} @{H} B={N}: {ms*1e3:.1f} us, {nwg} workgroups, tile h {th}, K chunks {nchunk}; median wave lifetime "
          f"{np.median(life):.0f} ticks; kernel span of the sampled half {span:.0f} ticks  (s_memtime ticks, 100 MHz on gfx9: "
          f"1 tick = 10 ns)")
    for (a, b), name in zip(SL, PH):
        if nchunk == 1 and 6 <= a <= 10 and b <= 11: continue
        if not stats_on and b == 13: continue
        d = t[:, :, b] - t[:, :, a]
        d = d[(t[:, :, b] > 0) & (t[:, :, a] > 0)]
        if d.size == 0: continue
        print(f"   {name:36s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  p90 {np.percentile(d, 90):8.0f}   "
              f"{100 * d.mean() / life.mean():5.1f} % of lifetime")
    # concurrency: average number of workgroups alive at a time on the chip
    starts, ends = t[:, :, 0].min(1), t[:, :, 12].max(1)
    alive = (ends - starts).sum() / max(span, 1)
    print(f"   average workgroups alive (sampled half) {alive:.0f}  -> about {2 * alive / 256:.1f} per CU")


for shape in [(32, 256, 32, 32), (32, 256, 16, 32), (32, 128, 64, 64), (32, 512, 32, 16), (32, 64, 128, 128)]:
    run(*shape)
