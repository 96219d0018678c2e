"""Round 4: wave-specialised weight gradient (wgrad_ws.hip) against wgrad_kernel.h.
  1. stand-alone, per U-Net layer shape (bs 32): ms per launch and TFLOP/s, AMX_WGRAD_WS = 0 / 1, bit-identity of the rows;
  2. in-step, interleaved in-process A/B of the training step: AMX_WGRAD_WS = 0 / 1 and per class mask.
Usage: python tools/gpu_wgrad_ws_ab.py [standalone|step|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
r16 = lambda v: (v + 15) // 16 * 16
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def standalone():
    # (H, C0, C1, Cout): every plain-3x3 MFMA weight gradient of the bs-32 U-Net step
    shapes = [(512, 16, 16, 16), (256, 16, 0, 32), (256, 32, 0, 32), (256, 32, 32, 32), (128, 32, 0, 64),
              (128, 64, 0, 64), (128, 64, 64, 64), (64, 64, 0, 128), (64, 128, 0, 128)]
    N = 32
    for (H, C0, C1, Cout) in shapes:
        X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
        sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
        sc1 = torch.rand(C1, device=dev) + 0.5 if C1 else None; sh1 = torch.randn(C1, device=dev) if C1 else None
        dpre = torch.randn(N, H, H, Cout, device=dev)
        rows = L.load().amx_conv2d_wgrad_rows(N, H, H, C0 + C1, Cout, 9, 1)
        part = torch.empty(rows, 9, r16(C0 + C1), r16(Cout), device=dev)
        sp = L.stream_ptr(dpre)
        def go():
            L.call("amx_conv2d_wgrad", L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), L.ptr(sc1), L.ptr(sh1), C1,
                   L.ptr(dpre), Cout, L.ptr(part), N, H, H, Cout, 9, 1, sp)
        out, keep = {}, {}
        for rep in range(2):
            for ws in ("0", "1"):
                os.environ["AMX_WGRAD_WS"] = ws
                for _ in range(2): go()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6): go()
                e1.record(); torch.cuda.synchronize()
                out.setdefault(ws, []).append(e0.elapsed_time(e1) / 6)
                keep[ws] = part.clone()
        fl = 2.0 * N * H * H * (C0 + C1) * Cout * 9
        a, b = min(out["0"]), min(out["1"])
        print(f"wgrad {C0}+{C1}->{Cout} @{H}^2: general {a:.4f} ms {fl / a / 1e9:6.1f} TF/s | ws {b:.4f} ms {fl / b / 1e9:6.1f} TF/s "
              f"({fl / b / 1e9 / 157.3:.3f} of peak) | x{a / b:.3f} | bit-identical {torch.equal(keep['0'], keep['1'])}", flush=True)


def step():
    import atomai_amd as aoi
    rs = np.random.RandomState(0)
    X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
    m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
    m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
    # (AMX_WGRAD_WS, AMX_WGRAD_WS_MASK, AMX_WGRAD_ORDER, AMX_CONV_WS_DGRAD)
    cfgs = [("0", "7", "0", "3"), ("1", "7", "0", "3"), ("1", "7", "1", "3"), ("0", "7", "1", "3"), ("1", "7", "1", "7"),
            ("1", "7", "2", "7"), ("1", "3", "1", "3"), ("1", "3", "0", "3")]
    if len(sys.argv) > 2:
        cfgs = [tuple(c.split(",")) for c in sys.argv[2:]]
    res = {c: [] for c in cfgs}
    for rep in range(3):
        for c in cfgs:
            (os.environ["AMX_WGRAD_WS"], os.environ["AMX_WGRAD_WS_MASK"], os.environ["AMX_WGRAD_ORDER"],
             os.environ["AMX_CONV_WS_DGRAD"]) = c
            for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(10): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / 10 * 1e3)
    for k, v in res.items():
        print(f"WGRAD_WS={k[0]} MASK={k[1]} ORDER={k[2]} CONV_WS_DGRAD={k[3]}: step ms {['%.3f' % t for t in v]}  min {min(v):.3f}", flush=True)


if what in ("standalone", "all"):
    standalone()
if what in ("step", "all"):
    step()
