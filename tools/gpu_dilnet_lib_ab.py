"""dilnet forward (eval, 16 frames of 1024^2): interleaved in-process A/B of library builds.
   usage: python tools/gpu_dilnet_lib_ab.py name1 name2 ...   (lib/libatomai_amd_<name>.so; 'default' = the product build)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd import _lib as L
from atomai_amd.nets.fcnn import predict_proba
names = sys.argv[1:] or ["default", "alt"]
libs = {n: (L.load() if n == "default" else L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{n}.so")))) for n in names}
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net.cuda().eval()
x = torch.from_numpy(np.random.RandomState(0).rand(16, 1, 1024, 1024).astype(np.float32)).cuda()
res = {}
for rep in range(3):
    for n, lb in libs.items():
        L._lib = lb
        for _ in range(2): predict_proba(net, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): predict_proba(net, x)
        torch.cuda.synchronize()
        res.setdefault(n, []).append((time.perf_counter() - t0) / 5 / 16 * 1e3)
for k, v in res.items():
    ms = min(v)
    print(f"{k:12s}: ms/frame {['%.4f' % t for t in v]} min {ms:.4f} = {91.62e9 / ms / 1e9 / 157.3:.3f} of peak", flush=True)
