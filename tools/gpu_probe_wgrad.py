"""A/B of wgrad kernel variants (env AMX_WGRAD_TH) on the config-2 layer shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
def run(N, H, C0, C1, Cout, iters=10):
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    dpre = torch.randn(N, H, H, Cout, device=dev)
    rows = L.load().amx_conv2d_wgrad_rows(N, H, H, C0 + C1, Cout, 9, 1)
    ci_pad, co_pad = (C0 + C1 + 15) // 16 * 16, (Cout + 15) // 16 * 16
    part = torch.empty(rows, 9, ci_pad, co_pad, device=dev)
    def go():
        L.call("amx_conv2d_wgrad", L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), None, None, C1, L.ptr(dpre), Cout,
               L.ptr(part), N, H, H, Cout, 9, 1, L.stream_ptr(dpre))
    for _ in range(2): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return round(ms, 4), round(2.0 * N * H * H * (C0 + C1) * Cout * 9 / ms / 1e9 / 157.3, 3), rows
shapes = [(256, 16, 0, 32), (256, 32, 0, 32), (128, 32, 0, 64), (128, 64, 0, 64), (64, 64, 0, 128), (64, 128, 0, 128),
          (128, 64, 64, 64), (256, 32, 32, 32), (512, 16, 16, 16)]
res = {}
for th in os.environ.get("PROBE_TH", "8,4").split(","):
    os.environ["AMX_WGRAD_TH"] = th
    for sh_ in shapes:
        res[f"th{th}_{sh_}"] = run(32, *sh_)
for sh_ in shapes:
    print(sh_, {k.split('_(')[0]: v for k, v in res.items() if k.endswith(str(sh_))}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_wgrad.json", "w"), indent=1)
