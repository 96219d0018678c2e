"""A/B of conv kernel variants (env AMX_CONV_NT / AMX_CONV_DBUF) on the config-2 layer shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
def run(N, H, C0, C1, Cout, iters=20):
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    X0 = torch.randn(N, H, H, C0, device=dev); X1 = torch.randn(N, H, H, C1, device=dev) if C1 else None
    sc = torch.rand(C0, device=dev) + 0.5; sh = torch.randn(C0, device=dev)
    n = L.load().amx_pack_weights_size(Cout, C0, C1, 9, 0)
    wpk = torch.empty(n, device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, C0, C0, C1, C1, 9, 0, L.stream_ptr(w))
    cop = (Cout + 15) // 16 * 16
    bias = torch.zeros(Cout, device=dev); y = torch.empty(N, H, H, Cout, device=dev)
    stats = torch.empty(L.load().amx_conv2d_num_tiles(N, H, H, 4), 2, cop, device=dev)
    def go():
        L.call("amx_conv2d_fwd", L.ptr(X0), L.ptr(sc), L.ptr(sh), C0, L.ptr(X1), None, None, C1, L.ptr(wpk), L.ptr(bias),
               None, L.ptr(y), Cout, None, 0, L.ptr(stats), N, H, H, Cout, 9, 1, 0.01, L.stream_ptr(y))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * N * H * H * Cin * Cout * 9 / ms / 1e9 / 157.3
# forward shapes of the bs-32 U-Net step, then the dgrad-only shapes (Cin/Cout swapped)
shapes = [(256, 16, 0, 32), (256, 32, 0, 32), (128, 32, 0, 64), (128, 64, 0, 64), (64, 64, 0, 128), (64, 128, 0, 128),
          (128, 64, 64, 64), (256, 32, 32, 32), (512, 16, 16, 16), (512, 16, 0, 32), (256, 32, 0, 64), (128, 64, 0, 128),
          (256, 32, 0, 16), (128, 64, 0, 32), (64, 128, 0, 64)]
res = {}
for th in ("8", "16"):
    for nt in ("1", "2", "4"):
        os.environ["AMX_CONV_TH"] = th; os.environ["AMX_CONV_NT"] = nt
        for sh_ in shapes:
            if int(nt) * 16 > (sh_[3] + 15) // 16 * 16: continue
            ms, fr = run(32, *sh_)
            res[f"th{th}_nt{nt}_{sh_}"] = (round(ms, 4), round(fr, 3))
for sh_ in shapes:
    row = {k.split('_(')[0]: v for k, v in res.items() if k.endswith(str(sh_))}
    print(sh_, row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_variants.json", "w"), indent=1)
