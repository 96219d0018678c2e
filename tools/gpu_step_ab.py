"""In-process A/B of library / environment-selected kernel variants on whole workloads (dev tool):
     U-Net bs 32 512^2 training step (ms/step, min of 3 x 8 steps) and dilnet 1024^2 predict (device ms/frame).
   python tools/gpu_step_ab.py "AMX_CONV_PERSIST=0" "AMX_CONV_PERSIST=1" "AMX_CONV_PERSIST=2" ...
The library freezes its switches at the first launch (csrc/knobs.hip); a variant change re-reads them through
amx_knobs_reload, so variants interleave in ONE process (same clocks, same allocator state)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba

import ctypes
from atomai_amd import _lib
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in (sys.argv[1:] or ["", "lib=ref"])]
KEYS = sorted({k for v in variants for k in v if k != "lib"})
_libs = {"": _lib.load()}


def setenv(v):
    """`lib=<name>` selects lib/libatomai_amd_<name>.so (same ABI as the product library), other keys are env switches."""
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update({k: x for k, x in v.items() if k != "lib"})
    name = v.get("lib", "")
    if name not in _libs:
        _libs[name] = _lib._bind(ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f"libatomai_amd_{name}.so")))
    _lib._lib = _libs[name]
    _lib.reload_knobs()                   # every build carries its own switch table
    from atomai_amd import engine
    engine._pack_cache.store.clear()      # weight images are library-specific (layout of a partial last K chunk)
    engine.FUSE_HEAD = os.environ.get("AMX_FUSE_HEAD", "1") != "0"     # (python-level switch)
    engine.FUSE_POOL = os.environ.get("AMX_FUSE_POOL", "1") != "0"
    engine.FUSE_UPCONV = os.environ.get("AMX_FUSE_UPCONV", "1") != "0"
    engine.FIRST_WGRAD_MAIN = os.environ.get("AMX_FIRST_WGRAD_MAIN", "1") != "0"
    engine.DGRAD_SPLIT = os.environ.get("AMX_DGRAD_SPLIT", "1") != "0"
    engine.FUSE_POOL_WGRAD1 = os.environ.get("AMX_FUSE_POOL_WGRAD1", "1") != "0"
    engine.FUSE_PX_LOSS = os.environ.get("AMX_FUSE_PX_LOSS", "1") != "0"
    from atomai_amd.trainers import trainer as _tr
    _tr.FUSE_LOSS = engine.FUSE_PX_LOSS
    from atomai_amd.losses_metrics import losses
    losses.LOSS_BLOCKS[0] = int(os.environ.get("AMX_LOSS_BLOCKS", "4096"))


rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
torch.manual_seed(1)
dn, _ = aoi.nets.init_fcnn_model("dilnet", 1)
dn = dn.cuda().eval()
xf = torch.from_numpy(rs.rand(8, 1, 1024, 1024).astype(np.float32)).cuda()
res = {json.dumps(v): {"unet_ms": [], "dilnet_ms_per_frame": []} for v in variants}
ref_out = None
for v in variants:                      # the variants must agree on the full-size dilnet output
    setenv(v)
    o = predict_proba(dn, xf[:2]).float()
    if ref_out is None: ref_out = o
    res[json.dumps(v)]["dilnet_max_abs_diff_vs_first"] = float((o - ref_out).abs().max())
for rep in range(3):
    for v in variants:
        setenv(v)
        for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize()
        res[json.dumps(v)]["unet_ms"].append((time.perf_counter() - t0) / 8 * 1e3)
        for _ in range(2): predict_proba(dn, xf)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): predict_proba(dn, xf)
        torch.cuda.synchronize()
        res[json.dumps(v)]["dilnet_ms_per_frame"].append((time.perf_counter() - t0) / 4 / 8 * 1e3)
for k, v in res.items():
    print(f"{k:60s} unet step {min(v['unet_ms']):7.3f} ms ({['%.2f' % t for t in v['unet_ms']]})   "
          f"dilnet {min(v['dilnet_ms_per_frame']):6.3f} ms/frame = {91.62e9 / min(v['dilnet_ms_per_frame']) / 1e9:5.1f} TF"
          f"  (max |out - first variant| {v['dilnet_max_abs_diff_vs_first']:.1e})", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/step_ab.json", "w"), indent=1)
