import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
stack = rs.rand(256, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
ref = p.run(stack[:24], compute_coords=False, chunk_bytes=None)
p.chunk_bytes = 8 << 20                                   # 2 frames per chunk: exercises every pipeline stage
small = p.run(stack[:7], compute_coords=False)
assert np.array_equal(small, ref[:7]) or np.allclose(small, ref[:7], atol=1e-6), np.abs(small - ref[:7]).max()
p.chunk_bytes = 64 << 20
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = p.run(stack, compute_coords=False)
    dt = time.perf_counter() - t0
    print(f"rep{rep}: end-to-end {dt/256*1e3:.3f} ms/frame ({256/dt:.1f} frames/s)", flush=True)
assert np.allclose(out[:24], ref, atol=1e-6)
dec, coords = p.run(stack[:40], compute_coords=True)
print("coords ok", len(coords), sum(len(v) for v in coords.values()))
