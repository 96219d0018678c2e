"""Where does a wave of the rVAE decoder BACKWARD kernel spend its time?  (dev tool)
Uses lib/libatomai_amd_rprof.so (tools/build_variant_lib.sh rprof "-DAMX_RDEC_PROFILE" rdecoder).  Runs the config-4
training step (64x64 patches, bs 512, hidden 128 x 2 layers) and prints the share of a wave's lifetime per phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomai_amd import _lib as L
import atomai_amd as aoi

lib = L._bind(ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_rprof.so")))
raw = ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libatomai_amd_rprof.so"))
raw.amx_rdec_set_profile_buffer.argtypes = [ctypes.c_void_p]
L._lib = lib
B = 512
rs = np.random.RandomState(0)
X = rs.rand(B * 2, 64, 64).astype(np.float32)
m = aoi.models.rVAE((64, 64), latent_dim=2, seed=0)
m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
m.compile_trainer((X, None), None, batch_size=B)
x = torch.from_numpy(X[:B]).cuda()
prof = torch.zeros(B * 8 * 8, dtype=torch.int64, device="cuda")


def step():
    m.optim.zero_grad()
    elbo = m.forward_compute_elbo(x)
    (-elbo).backward()
    m.optim.step()
    return elbo.item()


for _ in range(3): step()
raw.amx_rdec_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
step()
torch.cuda.synchronize()
raw.amx_rdec_set_profile_buffer(None)
t = prof.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = t[t[:, 7] > 0]
PH = ["coordinate layer (16 tanh/lane, LDS) + barrier", "hidden layers forward (MFMA + tanh epilogue + barriers)", "output-layer backward (VALU loops, barriers)",
      "wgrad MFMAs (b32 LDS operands)", "dgrad MFMAs (weights from L2 + b128 LDS operands)", "elementwise after dgrad + barriers", "coordinate-layer backward (VALU loops, barriers)"]
life = t[:, 7]
print(f"rdecoder_bwd_kernel<128,64,2>: {t.shape[0]} waves, 64 tiles per wave, median lifetime {np.median(life):.0f} clocks = {np.median(life)/64:.0f} per tile")
for i, name in enumerate(PH):
    print(f"   {name:62s} {100 * t[:, i].sum() / life.sum():5.1f} %   {t[:, i].sum() / t.shape[0] / 64:8.0f} clocks per tile")
print(f"   (outside the tile loop: {100 * (1 - t[:, :7].sum() / life.sum()):.1f} %)")
