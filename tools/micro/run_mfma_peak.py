import ctypes, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_peak.hip"), "-o", so])
import torch  # noqa: maps the HIP runtime first
lib = ctypes.CDLL(so); lib.mfma_peak_tflops.restype = ctypes.c_double
for wgs, iters, reps in ((256, 20000, 5), (1024, 20000, 5), (1024, 100000, 10), (2048, 50000, 10)):
    t0 = time.time(); tf = lib.mfma_peak_tflops(wgs, iters, reps)
    print(f"wgs={wgs} iters={iters} reps={reps}: {tf:.1f} TFLOP/s fp32 MFMA ({tf / 157.3:.3f} of 157.3; implies {tf / 157.3 * 2.4:.2f} GHz) wall {time.time() - t0:.2f}s", flush=True)
