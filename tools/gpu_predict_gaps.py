"""Main-stream gaps between chunks of the product SegPredictor pipeline (events around forward_)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
stack = rs.rand(256, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:16], compute_coords=False)
orig = p.forward_
evs, host = [], []
def fwd(x):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record(); r = orig(x); e1.record(); host.append((t, time.perf_counter())); evs.append((e0, e1)); return r
p.forward_ = fwd
data = p.preprocess(stack, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = p.batch_predict(data, (256, 1024, 1024, 1), 256)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
gpu = [a.elapsed_time(b) for a, b in evs]
gaps = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)]
print(f"pipeline {dt/256*1e3:.3f} ms/frame; forward GPU ms per chunk {np.round(gpu, 1).tolist()}")
print("gaps ms", np.round(gaps, 1).tolist())
print("host enqueue start offsets ms", [round((h[0] - t0) * 1e3, 1) for h in host])
