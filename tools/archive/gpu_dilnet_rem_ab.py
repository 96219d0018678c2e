"""Round 4: dilnet forward (eval, 16 frames of 1024^2) with the remainder-column classes (AMX_CONV_REM=1: 28 / 52 columns,
4-wide blocks on v_mfma_f32_4x4x1) against the padded power-of-two plan (AMX_CONV_REM=0), interleaved in one process; and
the same for a dilnet TRAINING step (bs 8, 512^2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net.cuda().eval()
x = torch.from_numpy(np.random.RandomState(0).rand(16, 1, 1024, 1024).astype(np.float32)).cuda()
res, outs = {}, {}
for rep in range(3):
    for rem in ("0", "1"):
        os.environ["AMX_CONV_REM"] = rem
        for _ in range(2): y = predict_proba(net, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = predict_proba(net, x)
        torch.cuda.synchronize()
        res.setdefault(rem, []).append((time.perf_counter() - t0) / 5 / 16 * 1e3)
        outs[rem] = y.detach().cpu()
for k, v in res.items():
    ms = min(v)
    print(f"predict AMX_CONV_REM={k}: ms/frame {['%.4f' % t for t in v]} min {ms:.4f} -> {91.62e9 / ms / 1e9:.1f} TFLOP/s = {91.62e9 / ms / 1e9 / 157.3:.3f} of peak", flush=True)
print("max |prob diff| REM 1 vs 0:", float((outs["1"] - outs["0"]).abs().max()))

rs = np.random.RandomState(0)
X = rs.rand(16, 512, 512).astype(np.float32); y = (rs.rand(16, 512, 512) > 0.5).astype(np.float32)
m = aoi.models.Segmentor("dilnet", nb_classes=1, seed=1)
m.compile_trainer((X, y, X[:8], y[:8]), training_cycles=10, batch_size=8)
res = {}
for rep in range(3):
    for rem in ("0", "1"):
        os.environ["AMX_CONV_REM"] = rem
        for i in range(3): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.synchronize()
        res.setdefault(rem, []).append((time.perf_counter() - t0) / 8 * 1e3)
for k, v in res.items():
    print(f"train bs8 512^2 AMX_CONV_REM={k}: step ms {['%.3f' % t for t in v]} min {min(v):.3f}", flush=True)
