"""A/B: training step on the default stream vs on a high-priority stream (the weight-gradient side stream then has
the lower priority, so the data-gradient critical path is dispatched first)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
hi = torch.cuda.Stream(priority=-1)
res = {"default": [], "main_high": []}
def run(n):
    for i in range(n): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
for rep in range(3):
    for mode in res:
        torch.cuda.synchronize()
        ctx = torch.cuda.stream(hi) if mode == "main_high" else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            run(8)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(12)
            torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 12 * 1e3)
for k, v in res.items():
    print(f"{k}: step ms {['%.2f' % t for t in v]}  min {min(v):.2f}", flush=True)
