"""python tools/micro/run_mfma_lds.py  (on the GPU box; builds tools/micro/mfma_lds.so if missing)"""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_lds.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "mfma_lds.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_lds_tflops.restype = ctypes.c_double
lib.mfma_lds_tflops.argtypes = [ctypes.c_int] * 4
names = {0: "pure MFMA", 1: "reads before use", 2: "reads one burst ahead", 3: "reads before use + barrier/9",
         4: "one burst ahead + barrier/9"}
for mode in range(5):
    row = []
    for occ in (1, 2, 3, 4):
        tf = lib.mfma_lds_tflops(mode, occ, 400, 3)
        row.append(f"occ{occ} {tf:6.1f}")
    print(f"mode {mode} {names[mode]:30s} " + "  ".join(row) + "  TFLOP/s", flush=True)
lib.mfma_rand_tflops.restype = ctypes.c_double
lib.mfma_rand_tflops.argtypes = [ctypes.c_int] * 4
for zero in (0, 1):
    for iters, reps in ((20000, 3), (200000, 3)):
        tf = lib.mfma_rand_tflops(zero, 4, iters, reps)
        print(f"pure MFMA, {'zero' if zero else 'random'} operands, 4 wg/CU, {iters} iters x {reps}: {tf:6.1f} TFLOP/s "
              f"(effective clock {tf / 157.3 * 2.4:.2f} GHz)", flush=True)
