"""dilnet forward (16 frames of 1024x1024, eval, fused head / sums) under the 48 + 4 column class of the dilated layers
(conv_kernel.h REM) vs the 2 x 32 column plan, for library variants built with different occupancy bounds.
   python tools/gpu_nt3_ab.py [variant ...]     variants = suffixes of atomai_amd/lib/libatomai_amd_<v>.so"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
import atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba

torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net = net.cuda().eval()
x = torch.rand(16, 1, 1024, 1024, device="cuda")
base = L.load()
libs = {"default": base}
for v in sys.argv[1:]:
    libs[v] = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{v}.so")))
ref = None
for rep in range(2):
    for name, lib in libs.items():
        L._lib = lib
        for nt3 in ("0", "1"):
            os.environ["AMX_CONV_NT3"] = nt3
            for _ in range(2): y = predict_proba(net, x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): y = predict_proba(net, x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 / 16
            if ref is None: ref = y.clone()
            err = float((y - ref).abs().max())
            print(f"lib={name:10s} NT3={nt3}: {dt*1e3:.3f} ms/frame  {91.62e9/dt/1e12:.1f} TFLOP/s  frac {91.62e9/dt/1e12/157.3:.3f}  max|d| vs first {err:.2e}", flush=True)
L._lib = base
