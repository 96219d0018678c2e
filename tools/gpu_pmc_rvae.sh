#!/bin/bash
# SQ counters of the rVAE decoder kernels (config 4 step, tools/bench_extra.py rvae): two PMC passes with kernel trace only,
# summarised into profiles/r03_pmc_rvae.md — hardware-counted MFMA work (issued vs algorithmic FLOPs) and MFMA pipe busy.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
AMX_RVAE_NO_AB=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace -d /root/repo/gpurun_out/pmc_rvae -o pmc --output-format csv -- python /root/repo/tools/bench_extra.py rvae > /root/repo/gpurun_out/pmc_rvae.log 2>&1
AMX_RVAE_NO_AB=1 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d /root/repo/gpurun_out/pmc_rvae2 -o pmc --output-format csv -- python /root/repo/tools/bench_extra.py rvae > /root/repo/gpurun_out/pmc_rvae2.log 2>&1
cd /root/repo
python - <<'PY' | tee gpurun_out/pmc_rvae_summary.txt
import collections, csv, glob, re
def load(sub):
    path = glob.glob(f"gpurun_out/{sub}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(per.values())
a, b = load("pmc_rvae"), load("pmc_rvae2")
for kern in ("rdecoder_fwd_kernel", "rdecoder_bwd_kernel"):
    A = [d for d in a if kern in d[0]][-3:]; B = [d for d in b if kern in d[0]][-3:]
    if not A or not B: continue
    n = len(A)
    busy = sum(d[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in A) / n
    sqb = sum(d[1].get("SQ_BUSY_CYCLES", 0) for d in A) / n
    wave = sum(d[1].get("SQ_WAVE_CYCLES", 0) for d in A) / n
    ldsc = sum(d[1].get("SQ_LDS_BANK_CONFLICT", 0) for d in A) / n
    mops = sum(d[1].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) for d in B) / len(B)
    gui = sum(d[1].get("GRBM_GUI_ACTIVE", 0) for d in B) / len(B)
    ns = sum(d[2] for d in B) / len(B)
    name = re.sub(r"\(.*", "", A[-1][0]).replace("void ", "")
    print(f"{name}: {ns/1e6:.3f} ms; MFMA MOPS_F32 {mops:.3e} -> issued {mops*512/1e9:.1f} GFLOP = {mops*512/ns/1e3:.1f} TFLOP/s; "
          f"GRBM_GUI_ACTIVE {gui:.3e}; SQ_BUSY_CYCLES {sqb:.3e}; SQ_VALU_MFMA_BUSY_CYCLES {busy:.3e}; SQ_WAVE_CYCLES {wave:.3e}; LDS bank conflict cycles {ldsc:.3e}")
PY
