"""Device-only dilnet forward (8 frames of 1024x1024) under AMX_CONV_NT overrides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net = net.cuda().eval()
x = torch.rand(8, 1, 1024, 1024, device="cuda")
for nt in ("", "1", "2", "4"):
    if nt: os.environ["AMX_CONV_NT"] = nt
    else: os.environ.pop("AMX_CONV_NT", None)
    for _ in range(2): predict_proba(net, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): predict_proba(net, x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 / 8
    print(f"AMX_CONV_NT={nt or 'plan'}: {dt*1e3:.3f} ms/frame  {91.62e9/dt/1e12:.1f} TF", flush=True)
