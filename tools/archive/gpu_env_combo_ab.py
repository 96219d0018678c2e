"""In-process interleaved A/B of the U-Net training step (bs 32, 512^2) over combinations of launch-time env knobs
(and, with the pseudo-key LIB=<name>, of library builds).
  usage: python tools/gpu_env_combo_ab.py "A=1,B=2" "A=0,B=2" ...      (each argument = one configuration)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctypes
import atomai_amd as aoi
from atomai_amd import _lib as L
cfgs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
keys = sorted({k for c in cfgs for k in c})
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
res = [[] for _ in cfgs]
LIBS = {"default": L.load()}
for rep in range(3):
    for ci, c in enumerate(cfgs):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in c.items() if k != "LIB"})
        # pseudo-key LIB=<name>: lib/libatomai_amd_<name>.so instead of the product build (tools/build_variant_lib.sh)
        name = c.get("LIB", "default")
        if name not in LIBS:
            LIBS[name] = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), f"libatomai_amd_{name}.so")))
        L._lib = LIBS[name]
        for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize()
        res[ci].append((time.perf_counter() - t0) / 10 * 1e3)
for a, v in zip(sys.argv[1:], res):
    print(f"{a:60s} step ms {['%.3f' % t for t in v]}  min {min(v):.3f}", flush=True)
