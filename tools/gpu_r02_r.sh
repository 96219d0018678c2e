cd /root/repo
cat > /tmp/pred2.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, time, atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
stack = np.random.RandomState(0).rand(512, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:32], compute_coords=False)
for n in (64, 256, 512):
    ts = []
    for _ in range(3):
        t=time.perf_counter(); p.run(stack[:n], compute_coords=False); ts.append(time.perf_counter()-t)
    print("threads %d; %d frames: best %.3f s = %.1f frames/s; all %s" % (torch.get_num_threads(), n, min(ts), n/min(ts), ["%.3f" % t for t in ts]), flush=True)
PY
for t in 8 2 16 0; do echo "== AMX_PREDICT_HOST_THREADS=$t"; AMX_PREDICT_HOST_THREADS=$t timeout 300 python /tmp/pred2.py 2>&1 | grep frames; done
