"""tools/micro/mfma_lds.hip modes 7+: wave-specialised producer / consumer workgroups vs the pure MFMA loop."""
import ctypes, os, functools
print = functools.partial(print, flush=True)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_lds.so"))
for f in (lib.mfma_lds_tflops, lib.mfma_spec_tflops, lib.mfma_rand_tflops):
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_int] * 4
print("pure MFMA random operands 4 wg/CU: %.1f TFLOP/s" % lib.mfma_rand_tflops(0, 4, 20000, 3))
for occ in (1, 2, 4):
    print("mode 3 (reads + barrier, unspecialised) occ%d: %.1f" % (occ, lib.mfma_lds_tflops(3, occ, 2000, 3)))
names = {0: "4 consumer + 4 producer waves, 6 loads + 600 VALU / phase", 1: "8 + 8 waves", 2: "4 + 4, 1200 VALU", 3: "4 + 4, no VALU ballast"}
for v in (0, 1, 2, 3):
    for occ in ((1, 2) if v != 1 else (1,)):
        print("specialised %-58s %d wg/CU: %.1f TFLOP/s" % (names[v], occ, lib.mfma_spec_tflops(v, occ, 300, 3)))
