"""Isolated timing of the HBM-bound first-layer / head kernels of the U-Net step across library variants (dev tool).
   python tools/gpu_small_kernels_ab.py [lib names ...]      ("" = product is always included)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L

dev = torch.device("cuda:0")
N, H, W, C, Cs, K = 32, 512, 512, 16, 16, 3
torch.manual_seed(0)
x = torch.rand(N, 1, H, W, device=dev)
w1 = torch.randn(C, 1, 3, 3, device=dev); b1 = torch.randn(C, device=dev)
y = torch.empty(N, H, W, Cs, device=dev)
a = torch.randn(N, H, W, Cs, device=dev)
dpre = torch.randn(N, H, W, Cs, device=dev)
lib0 = L.load()
rows, rows_pix = lib0.amx_rows_for(N * H * W), lib0.amx_rows_pix(N * H * W)
stats = torch.empty(rows, 2, 16, device=dev)
part1 = torch.empty(rows, 10, Cs, device=dev)
sc = torch.rand(Cs, device=dev) + 0.5; sh = torch.randn(Cs, device=dev)
wp = torch.randn(K, C, device=dev); bp = torch.randn(K, device=dev)
logits = torch.empty(N, K, H, W, device=dev); dl = torch.randn(N, K, H, W, device=dev) * 1e-6
dxn = torch.empty(N, H, W, Cs, device=dev); ppart = torch.empty(rows, K, Cs, device=dev); ppartb = torch.empty(rows, K, device=dev)
tgt = torch.randint(0, K, (N, H, W), device=dev); cepart = torch.empty(4096, device=dev)
sp = L.stream_ptr(x)


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# first-layer kernels with / without the LDS-staged neighbourhoods (AMX_CONV1_LDS is read per call), U-Net and dilnet shapes
xd = torch.rand(16, 1, 1024, 1024, device=dev); wd = torch.randn(25, 1, 3, 3, device=dev); bd = torch.randn(25, device=dev)
yd = torch.empty(16, 1024, 1024, 28, device=dev)
rd, rpd = lib0.amx_rows_for(16 * 1024 * 1024), lib0.amx_rows_pix(16 * 1024 * 1024)
for rep in range(2):
    for mode in ("0", "1"):
        os.environ["AMX_CONV1_LDS"] = mode
        t = {}
        t["conv1_fwd"] = timed(lambda: L.call("amx_conv1_fwd", L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(y), L.ptr(stats), N, H, W, C, Cs, 1, 0.01, rows, rows_pix, 0.0, 1.0, sp))
        t["conv1_wgrad"] = timed(lambda: L.call("amx_conv1_wgrad", L.ptr(x), L.ptr(dpre), L.ptr(part1), N, H, W, Cs, 1, rows, rows_pix, sp))
        t["dilnet conv1_fwd eval 16x1024^2x28"] = timed(lambda: L.call("amx_conv1_fwd", L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(yd), None, 16, 1024, 1024, 25, 28, 1, 0.01, rd, rpd, 0.1, 0.8, sp))
        print(f"AMX_CONV1_LDS={mode}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in t.items()), flush=True)
os.environ.pop("AMX_CONV1_LDS", None)

libs = {"": lib0}
for name in sys.argv[1:]:
    libs[name] = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{name}.so")))
for rep in range(2):
    for name, lib in libs.items():
        L._lib = lib
        t = {}
        t["conv1_fwd"] = timed(lambda: L.call("amx_conv1_fwd", L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(y), L.ptr(stats), N, H, W, C, Cs, 1, 0.01, rows, rows_pix, 0.0, 1.0, sp))
        t["conv1_wgrad"] = timed(lambda: L.call("amx_conv1_wgrad", L.ptr(x), L.ptr(dpre), L.ptr(part1), N, H, W, Cs, 1, rows, rows_pix, sp))
        t["px_fwd"] = timed(lambda: L.call("amx_px_fwd", L.ptr(a), L.ptr(sc), L.ptr(sh), L.ptr(wp), L.ptr(bp), L.ptr(logits), N, H, W, C, Cs, K, 0, sp))
        t["px_bwd"] = timed(lambda: L.call("amx_px_bwd", L.ptr(dl), L.ptr(a), L.ptr(sc), L.ptr(sh), L.ptr(wp), L.ptr(dxn), L.ptr(ppart), L.ptr(ppartb), None, N, H, W, C, Cs, K, rows, rows_pix, sp))
        print(f"lib {name or 'product':8s}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in t.items()), flush=True)
L._lib = lib0
