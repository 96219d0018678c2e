import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch, atomai_amd as aoi
from atomai_amd.nets.gp import convFeatureExtractor
rs = np.random.RandomState(0)
N, p = 16384, 16
X = rs.rand(N, p * p).astype(np.float32)
y = np.stack([X.reshape(N, p, p)[:, 4:12, 4:12].mean((1, 2)), X[:, :32].mean(1)]).astype(np.float32)
for prec in ("single", "double"):
    m = aoi.models.dklGPR(p * p, embedim=2, precision=prec, seed=1)
    t0 = time.perf_counter(); m.fit(X, y, training_cycles=20, feature_extractor=convFeatureExtractor, print_loss=20); torch.cuda.synchronize()
    t1 = time.perf_counter()
    mean, var = m.predict(X, batch_size=4096); t2 = time.perf_counter()
    s = m.sample_from_posterior(X[:2048], num_samples=16); t3 = time.perf_counter()
    ts, idx = m.thompson(X[:2048]); t4 = time.perf_counter()
    print(prec, "fit20 %.3fs predict16384 %.3fs sample2048 %.3fs thompson %.3fs" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3),
          "loss", m.train_loss[0], m.train_loss[-1], "mean err", float(np.abs(mean - y).mean()), "var range", float(var.min()), float(var.max()),
          "grid v", m.gp_model.grid.version, "r", None, "peak GB", torch.cuda.max_memory_allocated() / 1e9)
