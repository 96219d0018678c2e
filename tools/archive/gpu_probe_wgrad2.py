"""Per-layer sweep of the wgrad split-K target (AMX_WGRAD_WGS) and tile height (AMX_WGRAD_TH) on the config-2
layer shapes, including the partial-sum reduction chain.  Output: gpurun_out/probe_wgrad2.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
r16 = lambda v: (v + 15) // 16 * 16

def run(N, H, C0, C1, Cout, taps, iters=8):
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    dpre = torch.randn(N, H, H, Cout, device=dev)
    rows = L.load().amx_conv2d_wgrad_rows(N, H, H, C0 + C1, Cout, taps, 1)
    ci_pad, co_pad = r16(C0 + C1), r16(Cout)
    part = torch.empty(rows, taps, ci_pad, co_pad, device=dev)
    dw = torch.empty(Cout, C0 + C1, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=dev)
    ncols = taps * ci_pad * co_pad
    nch = 32 if rows < 1024 else 128
    part2 = torch.empty(nch, ncols, device=dev)
    sp = L.stream_ptr(dpre)
    def go():
        L.call("amx_conv2d_wgrad", L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), None, None, C1, L.ptr(dpre), Cout,
               L.ptr(part), N, H, H, Cout, taps, 1, sp)
    def red():
        p, r = part, rows
        if r > 64:
            L.call("amx_reduce_rows_chunked", L.ptr(p), r, ncols, nch, L.ptr(part2), sp)
            p, r = part2, -(-r // -(-r // nch))
        L.call("amx_wgrad_reduce", L.ptr(p), r, taps, ci_pad, co_pad, C0, C0, C1, Cout, L.ptr(dw), sp)
    def timeit(f):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    a, b = timeit(go), timeit(red)
    tf = 2.0 * N * H * H * (C0 + C1) * Cout * taps / (a + b) / 1e9
    return dict(ms=round(a, 4), red_ms=round(b, 4), tf=round(tf, 1), rows=rows)

# (H, C0, C1, Cout, taps): every MFMA wgrad of the bs-32 U-Net step
shapes = [(512, 16, 16, 16, 9), (256, 16, 0, 32, 9), (256, 32, 0, 32, 9), (256, 32, 32, 32, 9), (128, 32, 0, 64, 9),
          (128, 64, 0, 64, 9), (128, 64, 64, 64, 9), (64, 64, 0, 128, 9), (64, 128, 0, 128, 9),
          (64, 128, 0, 64, 1), (128, 64, 0, 32, 1), (256, 32, 0, 16, 1)]
res = {}
for sh_ in shapes:
    for th in ("4", "8"):
        if sh_[4] == 1 and th == "4":
            continue
        for wgs in ("256", "512", "1024", "2048"):
            os.environ["AMX_WGRAD_TH"] = th; os.environ["AMX_WGRAD_WGS"] = wgs
            res[f"{sh_}|th{th}|wg{wgs}"] = run(32, *sh_)
    best = max((v["tf"], k) for k, v in res.items() if k.startswith(str(sh_)))
    print(sh_, "best", best, {k.split('|', 1)[1]: (v["ms"], v["red_ms"]) for k, v in res.items() if k.startswith(str(sh_))}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_wgrad2.json", "w"), indent=1)
