cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
cat > /tmp/pred.py <<'PY'
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
stack = np.random.RandomState(0).rand(192, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:16], compute_coords=False)
p.run(stack, compute_coords=False)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /root/repo/gpurun_out/r02p_prof -o pred -- python /tmp/pred.py ) > gpurun_out/r02p_rocprof.log 2>&1
python tools/gpu_predict_timeline.py $(ls gpurun_out/r02p_prof/*.db | head -1) 2>&1 | tee gpurun_out/r02_predict_timeline_final.txt
rm -rf gpurun_out/r02p_prof
