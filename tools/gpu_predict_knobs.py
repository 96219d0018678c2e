"""Chunk size / chunks in flight of SegPredictor's pipeline on BASELINE configs[2] (dilnet, 1024^2 frames): end-to-end
frames/s of a 2048-frame call per (AMX_PREDICT_CHUNK_MB, AMX_PREDICT_NS) pair, interleaved, 2 repetitions (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi

frames, hw = 2048, 1024
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
base = torch.from_numpy(rs.rand(64, hw, hw).astype(np.float32))
stack = np.empty((frames, hw, hw), dtype=np.float32)
st = torch.from_numpy(stack)
for i in range(0, frames, 64):
    torch.mul(base, 0.5 + (i // 64) / 128.0, out=st[i:i + 64])
pairs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(64, 3), (128, 3), (128, 2), (256, 2), (32, 4), (64, 4)]
res = {p: [] for p in pairs}
ref = None
for rep in range(2):
    for mb, ns in pairs:
        os.environ["AMX_PREDICT_NS"] = str(ns)
        p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False, chunk_bytes=mb << 20)
        p.run(stack[:64], compute_coords=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = p.run(stack, compute_coords=False)
        dt = time.perf_counter() - t0
        res[(mb, ns)].append(frames / dt)
        chk = out[::129].copy()
        if ref is None:
            ref = chk
        assert np.array_equal(chk, ref), "chunking must not change the result"
        del out, p
for k, v in res.items():
    print(f"chunk {k[0]:4d} MB, {k[1]} in flight: {max(v):7.1f} frames/s  ({['%.1f' % x for x in v]})", flush=True)
