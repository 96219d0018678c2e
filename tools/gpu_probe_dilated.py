"""Per-layer timing of the dilnet convolutions (N=8 frames of 1024^2 -> 512^2 feature maps)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
dev = torch.device("cuda:0")
r4 = lambda v: (v + 3) // 4 * 4
def run(N, H, Cin, Cout, dil, iters=10):
    Cs = r4(Cin); cos = r4(Cout)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    X0 = torch.randn(N, H, H, Cs, device=dev)
    sc = torch.rand(Cs, device=dev) + 0.5; sh = torch.randn(Cs, device=dev)
    n = L.load().amx_pack_weights_size(Cout, Cs, 0, 9, 0)
    wpk = torch.empty(n, device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, Cin, Cs, 0, 0, 9, 0, L.stream_ptr(w))
    bias = torch.zeros(Cout, device=dev); y = torch.empty(N, H, H, cos, device=dev)
    def go():
        L.call("amx_conv2d_fwd", L.ptr(X0), L.ptr(sc), L.ptr(sh), Cs, None, None, None, 0, L.ptr(wpk), L.ptr(bias),
               None, L.ptr(y), cos, None, 0, None, N, H, H, Cout, 9, dil, 0.01, L.stream_ptr(y))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return round(ms, 4), round(2.0 * N * H * H * Cin * Cout * 9 / ms / 1e9, 1)
for (Cin, Cout, dil) in [(25, 50, 2), (50, 50, 4), (50, 50, 6), (50, 50, 2), (50, 50, 1), (48, 48, 2), (64, 64, 2), (64, 64, 6), (48, 64, 6)]:
    print((Cin, Cout, dil), run(8, 512, Cin, Cout, dil), flush=True)
