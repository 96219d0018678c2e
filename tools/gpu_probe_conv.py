"""GPU probe (dev tool): parity + timing of amx_conv2d_fwd on the U-Net layer shapes (config 2)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from atomai_amd import _lib as L

dev = torch.device("cuda:0")
print("torch", torch.__version__, torch.cuda.get_device_name(0), flush=True)
maps = open("/proc/self/maps").read()
L.load()
maps = open("/proc/self/maps").read()
hips = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
print("libamdhip64 mapped:", hips, flush=True)

def r4(c): return (c + 3) // 4 * 4
def nhwc(x, Cs):
    o = torch.zeros(x.shape[0], x.shape[2], x.shape[3], Cs, device=x.device)
    o[..., :x.shape[1]] = x.permute(0, 2, 3, 1)
    return o.contiguous()

def run(N, H, W, C0, C1, Cout, taps, dil=1, slope=0.01, check=True, iters=10, stats_on=True):
    torch.manual_seed(0)
    Cin = C0 + C1; C0s, C1s, Cos = r4(C0), r4(C1), r4(Cout)
    k = 3 if taps == 9 else 1
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    x0 = torch.randn(N, C0, H, W, device=dev)
    x1 = torch.randn(N, C1, H, W, device=dev) if C1 else None
    sc = torch.rand(C0s, device=dev) + 0.5; sh = torch.randn(C0s, device=dev)
    X0 = nhwc(x0, C0s); X1 = nhwc(x1, C1s) if C1 else None
    n = L.load().amx_pack_weights_size(Cout, C0s, C1s, taps, 0)
    wpk = torch.empty(n, device=dev)
    L.call("amx_pack_weights", L.ptr(w), L.ptr(wpk), Cout, C0, C0s, C1, C1s, taps, 0, L.stream_ptr(w))
    cop = (Cout + 15) // 16 * 16
    bias = torch.zeros(cop, device=dev); bias[:Cout] = b
    y = torch.empty(N, H, W, Cos, device=dev)
    nt = L.load().amx_conv2d_num_tiles(N, H, W, 16)
    stats = torch.empty(nt, 2, cop, device=dev) if stats_on else None
    def go():
        L.call("amx_conv2d_fwd", L.ptr(X0), L.ptr(sc), L.ptr(sh), C0s, L.ptr(X1), None, None, C1s,
               L.ptr(wpk), L.ptr(bias), None, L.ptr(y), Cos, None, 0, L.ptr(stats),
               N, H, W, Cout, taps, dil, slope, L.stream_ptr(y))
    go(); torch.cuda.synchronize()
    err = None
    if check:
        xin = x0 * sc[:C0].view(1, -1, 1, 1) + sh[:C0].view(1, -1, 1, 1)
        if C1: xin = torch.cat([xin, x1], 1)
        ref = F.leaky_relu(F.conv2d(xin, w, b, padding=dil if taps == 9 else 0, dilation=dil if taps == 9 else 1), slope)
        got = y[..., :Cout].permute(0, 3, 1, 2)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        if stats_on:
            s = stats[:, 0, :Cout].double().sum(0) / (N * H * W)
            merr = (s - ref.double().mean((0, 2, 3))).abs().max().item()
        else: merr = None
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * N * H * W * Cin * Cout * taps
    tf = flops / ms / 1e9
    # MIOpen timing for context
    xin = torch.cat([x0, x1], 1) if C1 else x0
    for _ in range(2): F.conv2d(xin, w, b, padding=dil if taps == 9 else 0, dilation=dil if taps == 9 else 1)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): F.conv2d(xin, w, b, padding=dil if taps == 9 else 0, dilation=dil if taps == 9 else 1)
    e1.record(); torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / iters
    rec = dict(N=N, H=H, W=W, C0=C0, C1=C1, Cout=Cout, taps=taps, dil=dil, relerr=err, ms=ms, tflops=tf,
               frac_peak=tf / 157.3, miopen_ms=ms_t)
    print(json.dumps(rec), flush=True)
    return rec

out = []
# small correctness cases incl. odd sizes
out.append(run(2, 37, 29, 8, 8, 24, 9, 1))
out.append(run(1, 40, 40, 28, 28, 25, 9, 1))
out.append(run(1, 64, 64, 52, 0, 50, 9, 6))
out.append(run(2, 32, 32, 128, 0, 64, 1))
# config-2 layer shapes (B=32)
B = 32
for (H, C0, C1, Co) in [(256, 16, 0, 32), (256, 32, 0, 32), (128, 32, 0, 64), (128, 64, 0, 64), (64, 64, 0, 128),
                        (64, 128, 0, 128), (128, 64, 64, 64), (256, 32, 32, 32), (512, 16, 16, 16)]:
    out.append(run(B, H, H, C0, C1, Co, 9, 1))
out.append(run(B, 64, 64, 128, 0, 64, 1, slope=1.0, stats_on=False))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe_conv.json", "w"), indent=1)
