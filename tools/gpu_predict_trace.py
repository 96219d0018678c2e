import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
stack = rs.rand(512, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:64], compute_coords=False)
torch.cuda.synchronize()
os.environ["AMX_PREDICT_TRACE"] = "1"
t0 = time.perf_counter()
out = p.run(stack, compute_coords=False)
dt = time.perf_counter() - t0
print(f"512 frames in {dt:.3f} s = {512/dt:.1f} frames/s")
