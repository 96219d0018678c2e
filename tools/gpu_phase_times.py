import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.engine import Tape
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for side in (True,):
    Tape.use_side_stream = side
    for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
    torch.cuda.synchronize()
    tf = tb = to = tc = 0.0
    for i in range(6):
        m.net.train(); m.optimizer.zero_grad()
        c0 = time.perf_counter()
        e0 = ev(); out = m.net(m.X_train[i % 2]); loss = m.criterion(out, m.y_train[i % 2]); e1 = ev()
        loss.backward(); e2 = ev(); m.optimizer.step(); e3 = ev()
        c1 = time.perf_counter()
        torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2); to += e2.elapsed_time(e3); tc += (c1 - c0) * 1e3
    print(f"side_stream={side}: fwd {tf/6:.2f} ms  bwd {tb/6:.2f} ms  adam {to/6:.3f} ms  | host issue time {tc/6:.2f} ms/step", flush=True)
