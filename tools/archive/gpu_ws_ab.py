"""Per-layer A/B on the MI355X: the wave-specialised thin-layer kernel (conv_ws.hip, AMX_CONV_WS=1) against the general
kernel (AMX_CONV_WS=0) on the thin plain-3x3 shapes of the config-2 U-Net step (bs 32, 512^2): forward with statistics,
and the data-gradient form (no bias / statistics, two outputs).   python tools/gpu_ws_ab.py -> stdout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.load()


def run(tag, N, H, C0, C1, Cout, Y1=0, stats_on=True, iters=30):
    torch.manual_seed(0)
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) / ((C0 + C1) * 9) ** 0.5
    X0 = torch.randn(N, H, H, C0, device=dev)
    X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    wpk = torch.empty(lib.amx_pack_weights_size(Cout, C0, C1, 9, 0), device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, C0, C0, C1, C1, 9, 0, L.stream_ptr(w))
    bias = torch.randn(Cout, device=dev) if stats_on else None
    Y0 = Cout - Y1
    res = {}
    for ws in ("0", "1", "0", "1"):
        os.environ["AMX_CONV_WS"] = ws
        os.environ["AMX_CONV_WS_DGRAD"] = "7"          # every data-gradient class (the product default leaves 32 -> 16 out)
        y = torch.zeros(N, H, H, Y0, device=dev)
        y1 = torch.zeros(N, H, H, Y1, device=dev) if Y1 else None
        stats = None
        if stats_on:
            th = lib.amx_conv2d_tile_h(C0 + C1, Cout, 9, 1, H)
            stats = torch.zeros(lib.amx_conv2d_num_tiles(N, H, H, th), 2, Cout, device=dev)

        def launch():
            L.call("amx_conv2d_fwd", L.ptr(X0), L.ptr(sc) if stats_on else None, L.ptr(sh) if stats_on else None, C0, L.ptr(X1), None, None, C1,
                   L.ptr(wpk), L.ptr(bias), None, L.ptr(y), Y0, L.ptr(y1), Y1, L.ptr(stats), N, H, H, Cout, 9, 1,
                   0.01 if stats_on else 1.0, L.stream_ptr(y))
        n0 = lib.amx_conv2d_ws_launches()
        for _ in range(3): launch()
        assert (lib.amx_conv2d_ws_launches() - n0 == 3) == (ws == "1")
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * N * H * H * (C0 + C1) * Cout * 9
        res.setdefault(ws, []).append(ms)
        res["y" + ws] = (y.clone(), None if y1 is None else y1.clone(), None if stats is None else stats.clone())
        print(f"{tag:34s} WS={ws}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s  frac {fl / ms / 1e9 / 157.3:.3f}", flush=True)
    d = float((res["y1"][0] - res["y0"][0]).abs().max())
    d1 = 0.0 if Y1 == 0 else float((res["y1"][1] - res["y0"][1]).abs().max())
    ds = 0.0 if not stats_on else float(((res["y1"][2] - res["y0"][2]).abs() / (res["y0"][2].abs() + 1.0)).max())
    print(f"{'':34s} max|dy| {d:.2e} / {d1:.2e}   max rel d(stats) {ds:.2e}   speed-up {min(res['0']) / min(res['1']):.3f}", flush=True)


run("c2a fwd 16->32 @256", 32, 256, 16, 0, 32)
run("c2b/c5b fwd 32->32 @256", 32, 256, 32, 0, 32)
run("c6 fwd 16+16->16 @512", 32, 512, 16, 16, 16)
run("c6 dgrad 16->16|16 @512", 32, 512, 16, 0, 32, Y1=16, stats_on=False)
run("c2b/c5b dgrad 32->32 @256", 32, 256, 32, 0, 32, stats_on=False)
run("c2a dgrad 32->16 @256", 32, 256, 32, 0, 16, stats_on=False)
