"""Host (launch) time vs device time of one training step: how close is the step to being launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
for i in range(4): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
torch.cuda.synchronize()
host, total = [], []
for i in range(10):
    feat, tar = m.X_train[i % 2], m.y_train[i % 2]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.net.train(); m.optimizer.zero_grad()
    prob = m.net(feat); t1 = time.perf_counter()
    loss = m.criterion(prob, tar); loss.backward(); t2 = time.perf_counter()
    m.optimizer.step(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    host.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3)); total.append((t4 - t0) * 1e3)
h = np.array(host)
print("host enqueue ms: fwd %.2f  loss+bwd %.2f  adam %.2f  | sum %.2f ; step wall %.2f (min %.2f)" % (
    h[:, 0].mean(), h[:, 1].mean(), h[:, 2].mean(), h.sum(1).mean(), np.mean(total), np.min(total)))
print("per-step wall", ["%.1f" % t for t in total]); print("per-step host", ["%.1f/%.1f" % (a, b) for a, b, c in host])
print("cpu count", os.cpu_count(), "loadavg", os.getloadavg())
