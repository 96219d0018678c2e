"""Where does the end-to-end predict time go?  (host preprocessing vs the streamed device pipeline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
frames = 64
stack = rs.rand(frames, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:8], compute_coords=False)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = p.preprocess(stack, True); t1 = time.perf_counter()
    out = p.batch_predict(x, (frames, 1024, 1024, 1), frames); torch.cuda.synchronize(); t2 = time.perf_counter()
    o = out.numpy(); t3 = time.perf_counter()
    print(f"rep{rep}: preprocess {1e3*(t1-t0)/frames:.3f} ms/frame, pipeline {1e3*(t2-t1)/frames:.3f} ms/frame, total {1e3*(t3-t0)/frames:.3f}")
t0 = time.perf_counter(); mn = stack.min(); mx = stack.max(); t1 = time.perf_counter()
print(f"numpy min+max: {1e3*(t1-t0)/frames:.3f} ms/frame ({stack.nbytes*2/(t1-t0)/1e9:.1f} GB/s)")
pin = torch.empty((16, 1024, 1024), pin_memory=True)
t0 = time.perf_counter()
for s in range(0, frames, 16): pin.copy_(torch.from_numpy(stack[s:s+16]))
t1 = time.perf_counter()
print(f"staging memcpy: {1e3*(t1-t0)/frames:.3f} ms/frame ({stack.nbytes/(t1-t0)/1e9:.1f} GB/s)")
d = torch.empty((16, 1024, 1024), device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(4): d.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"H2D from pinned: {pin.numel()*4*4/(t1-t0)/1e9:.1f} GB/s")
