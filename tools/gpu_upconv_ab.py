"""UpsampleBlock forward: the one-pass kernel (upconv.hip) against amx_conv2d_fwd(taps = 1) + amx_upsample2x_fwd on the
UpsampleBlock shapes of BASELINE configs 2 and 3 (dev tool): microseconds per call (HIP events, 20 calls), HBM-level GB/s
of the algorithmic bytes (input read once + output written once), bit-identity of the outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomai_amd import _lib as L
from atomai_amd import engine as E

import ctypes
libs = {"product": L.load()}
for nm in sys.argv[1:]:
    libs[nm] = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{nm}.so")))


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, cin, cout, N, h, w, affine in (("dilnet up1", 50, 25, 16, 512, 512, False), ("unet up1", 128, 64, 32, 64, 64, True),
                                         ("unet up2", 64, 32, 32, 128, 128, True), ("unet up3", 32, 16, 32, 256, 256, True)):
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(cin, cout, 1).cuda()
    x = torch.randn(N, cin, h, w, device="cuda")
    tape = E.Tape(False, False)
    src = tape.input(x).out
    cs_in, cs_out = src.Cs, E.r4(cout)
    if affine:
        src.scale, src.shift = E.padded_vec(torch.rand(cin, device="cuda") + 0.5, cs_in), E.padded_vec(torch.randn(cin, device="cuda"), cs_in)
    def two():
        t = E.Tape(False, False)
        v = t.conv([src], conv, None, 1.0)
        return t.upsample(v, "bilinear").t
    y = torch.empty((N, 2 * h, 2 * w, cs_out), dtype=torch.float32, device="cuda")
    wt, bs = conv.weight.detach().contiguous(), conv.bias.detach()
    def one():
        L.call("amx_upconv1x1_fwd", L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(wt), L.ptr(bs), L.ptr(y), N, h, w,
               cin, cs_in, cout, cs_out, 0, L.stream_ptr(y))
        return y
    ref = two()
    t2 = timeit(two)
    gb = (N * h * w * cs_in + N * 4 * h * w * cs_out) * 4 / 1e9
    for nm, lib in libs.items():
        L._lib = lib
        got = one().clone()
        same = torch.equal(ref[..., :cout], got[..., :cout])
        t1 = timeit(one)
        print(f"{name:11s} {cin:3d}->{cout:2d} @{h}^2 x{N}: two kernels {t2:7.1f} us (incl. host glue), one pass [{nm}] {t1:7.1f} us = "
              f"{gb / t1 * 1e6:6.0f} GB/s of {gb:.2f} GB; bit-identical {same}", flush=True)
    L._lib = libs["product"]
