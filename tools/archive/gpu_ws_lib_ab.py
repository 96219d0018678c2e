"""conv_ws.hip: interleaved in-process A/B of two BUILDS (default vs lib/libatomai_amd_alt.so) on the thin-layer shapes of the
U-Net step (bs 32), forward (statistics on) and data-gradient form; plus the training step with both builds."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
libs = {"default": L.load(), "alt": L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_alt.so")))}
os.environ["AMX_CONV_WS"] = "1"; os.environ["AMX_CONV_WS_DGRAD"] = "7"


def run(tag, N, H, C0, C1, Cout, Y1=0, stats_on=True, iters=30):
    torch.manual_seed(0)
    lib = libs["default"]
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) / ((C0 + C1) * 9) ** 0.5
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    wpk = torch.empty(lib.amx_pack_weights_size(Cout, C0, C1, 9, 0), device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, C0, C0, C1, C1, 9, 0, L.stream_ptr(w))
    bias = torch.randn(Cout, device=dev) if stats_on else None
    Y0 = Cout - Y1
    y = torch.zeros(N, H, H, Y0, device=dev); y1 = torch.zeros(N, H, H, Y1, device=dev) if Y1 else None
    stats = None
    if stats_on:
        th = lib.amx_conv2d_tile_h(C0 + C1, Cout, 9, 1, H)
        stats = torch.zeros(lib.amx_conv2d_num_tiles(N, H, H, th), 2, Cout, device=dev)
    res, keep = {}, {}
    for rep in range(3):
        for k, lb in libs.items():
            L._lib = lb
            def launch():
                L.call("amx_conv2d_fwd", L.ptr(X0), L.ptr(sc) if stats_on else None, L.ptr(sh) if stats_on else None, C0, L.ptr(X1), None, None, C1,
                       L.ptr(wpk), L.ptr(bias), None, L.ptr(y), Y0, L.ptr(y1), Y1, L.ptr(stats), N, H, H, Cout, 9, 1,
                       0.01 if stats_on else 1.0, L.stream_ptr(y))
            for _ in range(3): launch()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): launch()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(k, []).append(e0.elapsed_time(e1) / iters)
            keep[k] = y.clone()
    fl = 2.0 * N * H * H * (C0 + C1) * Cout * 9
    a, b = min(res["alt"]), min(res["default"])
    print(f"{tag:30s} alt {a * 1e3:7.1f} us ({fl / a / 1e9 / 157.3:.3f})  default {b * 1e3:7.1f} us ({fl / b / 1e9 / 157.3:.3f})  x{a / b:.3f}  "
          f"bit-identical {torch.equal(keep['alt'], keep['default'])}", flush=True)


run("c2a fwd 16->32 @256", 32, 256, 16, 0, 32)
run("c2b/c5b fwd 32->32 @256", 32, 256, 32, 0, 32)
run("c6 fwd 16+16->16 @512", 32, 512, 16, 16, 16)
run("c6 dgrad 16->16|16 @512", 32, 512, 16, 0, 32, Y1=16, stats_on=False)
run("c2b/c5b dgrad 32->32 @256", 32, 256, 32, 0, 32, stats_on=False)
run("c2a dgrad 32->16 @256", 32, 256, 32, 0, 16, stats_on=False)

import atomai_amd as aoi
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); yy = rs.randint(0, 3, (64, 512, 512))
os.environ.pop("AMX_CONV_WS_DGRAD")
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, yy, X[:32], yy[:32]), training_cycles=10, batch_size=32)
res = {k: [] for k in libs}
for rep in range(3):
    for k, lb in libs.items():
        L._lib = lb
        for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 10 * 1e3)
for k, v in res.items():
    print(f"{k}: step ms {['%.3f' % t for t in v]}  min {min(v):.3f}", flush=True)
