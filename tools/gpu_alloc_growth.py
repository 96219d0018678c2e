"""Does the caching allocator keep growing in steady state?  (reserved bytes / hipMalloc calls every 10 steps)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
for blk in range(8):
    t0 = time.perf_counter()
    for i in range(10): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
    torch.cuda.synchronize()
    s = torch.cuda.memory_stats()
    print(f"steps {10*(blk+1):3d}: {1e2*(time.perf_counter()-t0):.2f} ms/step  reserved {s['reserved_bytes.all.current']/1e9:.2f} GB  "
          f"allocated {s['allocated_bytes.all.current']/1e9:.2f} GB  peak {s['allocated_bytes.all.peak']/1e9:.2f}  "
          f"hipMalloc {s['num_device_alloc']}  retries {s['num_alloc_retries']}  segments {s['segment.all.current']}", flush=True)
