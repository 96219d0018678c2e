import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd import engine as E
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
modes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,2,10,15").split(",")]
res = {k: [] for k in modes}
for rep in range(3):
    for mode in modes:
        E.FUSE = mode
        for i in range(2): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 8 * 1e3)
for k, v in res.items():
    print(f"FUSE={k:2d}: step ms {['%.2f' % t for t in v]}  min {min(v):.2f}", flush=True)
