"""Round-2 convolution A/B on the MI355X (dev tool): per-layer-shape timing of amx_conv2d_fwd for every library
lib/libatomai_amd*.so found next to the product library:
  ref        the round-1 library (tools/build_ref_lib.sh)
  cur        the working tree's product library
  <name>     compile-time variants (tools/build_variant_lib.sh <name> "<-D flags>" conv_fwd_3x3 ...)
on the config-2 (U-Net bs 32, 512^2) forward + data-gradient shapes and the config-3 (dilnet 1024^2) shapes.
Every library's output is checked against the first one's (bit-exact: same fp32 FMA order).
   python tools/gpu_probe_r02.py [unet] [dilnet]  ->  gpurun_out/r02_probe_conv.json
"""
import ctypes, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L

dev = torch.device("cuda:0")
PEAK = 157.3
new = L.load()
LIBS = {}
for path in sorted(glob.glob(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_*.so"))):
    name = os.path.basename(path)[len("libatomai_amd_"):-3]
    lib = ctypes.CDLL(path)
    for fn_name in ("amx_conv2d_fwd", "amx_pack_weights", "amx_pack_weights_size", "amx_conv2d_num_tiles", "amx_conv2d_tile_h"):
        fn = getattr(lib, fn_name)
        fn.restype, fn.argtypes = L.SIGNATURES[fn_name]
    LIBS[name] = lib
LIBS = {**({"ref": LIBS.pop("ref")} if "ref" in LIBS else {}), "cur": new, **LIBS}


def r4(c): return (c + 3) // 4 * 4
def r16(c): return (c + 15) // 16 * 16


def setup(N, H, C0, C1, Cout, taps, dil, stats_on):
    torch.manual_seed(0)
    C0s, C1s, Cos = r4(C0), r4(C1), r4(Cout)
    k = 3 if taps == 9 else 1
    w = torch.randn(Cout, C0 + C1, k, k, device=dev) / ((C0 + C1) * k * k) ** 0.5
    X0 = torch.randn(N, H, H, C0s, device=dev); X0[..., C0:] = 0
    X1 = None
    if C1:
        X1 = torch.randn(N, H, H, C1s, device=dev); X1[..., C1:] = 0
    sc = torch.rand(C0s, device=dev) + 0.5; sh = torch.randn(C0s, device=dev)
    n = new.amx_pack_weights_size(Cout, C0s, C1s, taps, 0)
    wpk = torch.empty(n, device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, C0, C0s, C1, C1s, taps, 0, L.stream_ptr(w))
    bias = torch.randn(r16(Cout), device=dev)
    return dict(N=N, H=H, C0s=C0s, C1s=C1s, Cos=Cos, Cout=Cout, taps=taps, dil=dil, X0=X0, X1=X1, sc=sc, sh=sh,
                wpk=wpk, bias=bias, stats_on=stats_on)


def launch(lib, S, y, stats):
    rc = lib.amx_conv2d_fwd(L.ptr(S["X0"]), L.ptr(S["sc"]), L.ptr(S["sh"]), S["C0s"], L.ptr(S["X1"]), None, None, S["C1s"],
                            L.ptr(S["wpk"]), L.ptr(S["bias"]), None, L.ptr(y), S["Cos"], None, 0, L.ptr(stats),
                            S["N"], S["H"], S["H"], S["Cout"], S["taps"], S["dil"], 0.01, L.stream_ptr(y))
    assert rc == 0, (rc, L.last_error())


def bufs(lib, S):
    y = torch.empty(S["N"], S["H"], S["H"], S["Cos"], device=dev)
    stats = None
    if S["stats_on"]:
        th = lib.amx_conv2d_tile_h(S["C0s"] + S["C1s"], S["Cout"], S["taps"], S["dil"], S["H"])
        stats = torch.zeros(lib.amx_conv2d_num_tiles(S["N"], S["H"], S["H"], th), 2, r16(S["Cout"]), device=dev)
    return y, stats


def timeit(lib, S, iters):
    y, stats = bufs(lib, S)
    for _ in range(3): launch(lib, S, y, stats)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): launch(lib, S, y, stats)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, y, stats


def probe(tag, N, H, C0, C1, Cout, taps=9, dil=1, stats_on=True, iters=20):
    S = setup(N, H, C0, C1, Cout, taps, dil, stats_on)
    fl = 2.0 * N * H * H * (C0 + C1) * Cout * taps
    row = {}
    y0 = s0 = None
    for name, lib in LIBS.items():
        ms, y, st = timeit(lib, S, iters)
        row[name] = ms
        if y0 is None:
            y0, s0 = y, st
        elif name.startswith("x_"):
            pass                                             # timing-only experiment libraries: results are not valid
        else:
            assert torch.equal(y, y0), (tag, name, float((y - y0).abs().max()))
            if st is not None and s0 is not None and st.shape == s0.shape:
                assert torch.allclose(st, s0, rtol=1e-5, atol=1e-4), (tag, name, "stats")
    best = min(row, key=row.get)
    print(f"{tag:30s} " + "  ".join(f"{k} {v*1e3:7.1f}us {fl/v/1e9/PEAK:5.3f}" for k, v in row.items()) + f"   best {best}", flush=True)
    return {"shape": tag, "gflop": fl / 1e9, **{k: round(v, 5) for k, v in row.items()}}


out = []
what = sys.argv[1:] or ["unet", "dilnet"]
if "unet" in what:
    B = 32
    print("== U-Net bs 32 512^2: forward shapes (stats on), then data-gradient shapes (no stats)")
    for (H, C0, C1, Co) in [(256, 16, 0, 32), (256, 32, 0, 32), (128, 32, 0, 64), (128, 64, 0, 64), (64, 64, 0, 128),
                            (64, 128, 0, 128), (128, 64, 64, 64), (256, 32, 32, 32), (512, 16, 16, 16)]:
        out.append(probe(f"fwd {C0}+{C1}->{Co} @{H}", B, H, C0, C1, Co))
    for (H, C0, Co) in [(256, 32, 16), (256, 32, 32), (128, 64, 32), (128, 64, 64), (64, 128, 64), (64, 128, 128),
                        (128, 64, 128), (256, 32, 64), (512, 16, 32)]:
        out.append(probe(f"dgrad {C0}->{Co} @{H}", B, H, C0, 0, Co, stats_on=False))
    for (H, C0, Co) in [(64, 128, 64), (128, 64, 32), (256, 32, 16)]:
        out.append(probe(f"1x1 {C0}->{Co} @{H}", B, H, C0, 0, Co, taps=1, stats_on=False))
if "dilnet" in what:
    print("== dilnet 1024^2, 4 frames (eval: no stats)")
    B = 4
    for (H, C0, C1, Co, d) in [(512, 25, 0, 50, 2), (512, 50, 0, 50, 4), (512, 50, 0, 50, 6), (512, 50, 0, 50, 2),
                               (1024, 25, 25, 25, 1), (512, 50, 0, 25, 0)]:
        taps = 9 if d else 1
        d = max(d, 1)
        out.append(probe(f"dil{d} {C0}+{C1}->{Co} @{H}", B, H, C0, C1, Co, taps, d, stats_on=False, iters=10))
if "occ" in what:
    print("== occupancy sweep (AMX_CONV_MAXWG = max workgroups per CU through an LDS pad), product library only")
    LIBS = {"cur": new}
    for (H, C0, C1, Co) in [(256, 32, 0, 32), (512, 16, 16, 16), (128, 64, 0, 64), (64, 128, 0, 128)]:
        for k in ("1", "2", "3", "4", "5", ""):
            if k: os.environ["AMX_CONV_MAXWG"] = k
            else: os.environ.pop("AMX_CONV_MAXWG", None)
            out.append(probe(f"maxwg={k or 'none'} {C0}+{C1}->{Co} @{H}", 32, H, C0, C1, Co))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_probe_conv.json", "w"), indent=1)
