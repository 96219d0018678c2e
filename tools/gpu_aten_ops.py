"""Which ATen operators (copies, fills, elementwise) still run inside one U-Net training step, and from where:
torch.profiler over 3 steady steps, CUDA-time per op name with the innermost atomai_amd / bench frame of its stack (dev tool)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from torch.profiler import profile, ProfilerActivity

rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
for i in range(4): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    dev_us = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    where = next((f for f in (e.stack or []) if "atomai_amd" in f or "bench" in f), "?")
    agg[(e.name, where.split("repo/")[-1][:90])][0] += 1
    agg[(e.name, where.split("repo/")[-1][:90])][1] += dev_us
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{n / 3:6.1f}/step {us / 3:8.1f} us/step  {name:28s} {where}")
