"""Where does the END-TO-END predict time go (config 3: dilnet over a stack of 1024^2 frames)?  (dev tool)
Device compute is ~1.4 ms/frame; everything else is host work on one Python thread: min / ptp of the stack, staging
copies, the output array.  Prints frames/s for a few stack sizes and a cProfile of one run."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.predictors.predictor import _min_ptp

torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
stack = rs.rand(256, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:16], compute_coords=False)
print("torch threads", torch.get_num_threads(), flush=True)
for _ in range(2):
    t0 = time.perf_counter(); m = _min_ptp(stack); t1 = time.perf_counter()
    n0 = time.perf_counter(); m2 = (stack.min(), np.ptp(stack)); n1 = time.perf_counter()
    print(f"min/ptp of 1 GB: torch.aminmax {1e3*(t1-t0):.1f} ms, numpy min + ptp {1e3*(n1-n0):.1f} ms, equal {m == m2}", flush=True)
for n in (64, 256, 256):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = p.run(stack[:n], compute_coords=False)
    dt = time.perf_counter() - t0
    print(f"{n} frames: {dt:.3f} s = {n/dt:.1f} frames/s ({1e3*dt/n:.2f} ms/frame)", flush=True)
pr = cProfile.Profile(); pr.enable()
p.run(stack, compute_coords=False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
