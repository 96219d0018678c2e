"""Summary of tools/gpu_pmc_dilnet.sh (gpurun_out/pmc_dilnet) -> profiles/<round>_pmc_dilnet.md (python tools/summarize_pmc_dilnet.py r03)."""
import sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
import csv, glob, collections, re
path = glob.glob("gpurun_out/pmc_dilnet/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    d = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"])])
    d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = list(per.values())
last = [i for i, d in enumerate(disp) if "conv1_fwd_kernel" in d[0]][-1]
disp = disp[last:]
tot_ns = sum(d[2] for d in disp); tot_mops = sum(d[1].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) for d in disp)
ALG = 91.62e9 * 16
lines = ["# Hardware-counted MFMA work of one dilnet forward (16 frames of 1024², eval, fused head / sums), MI355X, " + RND + "", "",
         "`tools/gpu_pmc_dilnet.sh`: `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace`.", "",
         "| kernel | ms | issued GFLOP (MOPS x 512) | issued TFLOP/s | MFMA busy per SIMD-cycle |", "|---|---|---|---|---|"]
for name, c, ns, grid in disp:
    m = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) * 512
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 0) * 128.0, 1)
    lines.append(f"| `{re.sub(r'[(].*', '', name).replace('void ', '')[:58]}` | {ns/1e6:.3f} | {m/1e9:.1f} | {m/ns/1e3:.1f} | {busy:.3f} |")
lines += ["", f"Forward: {tot_ns/1e6:.2f} ms of kernels; issued MFMA work {tot_mops*512/1e12:.3f} TFLOP vs {ALG/1e12:.3f} TFLOP algorithmic "
          f"(91.62 GFLOP per frame) = {tot_mops*512/ALG:.3f}x: {100*(1-ALG/(tot_mops*512)):.1f} % of the issued matrix work is channel padding "
          f"(r03: 25 / 50 filters in 32 / 64 MFMA columns; r04: 28 / 52 columns on the dilated layers, the head-fused last layer still 32; the K direction is exact).  Issued rate {tot_mops*512/tot_ns/1e3:.1f} TFLOP/s, "
          f"algorithmic {ALG/tot_ns/1e3:.1f} TFLOP/s over the whole forward."]
open(f"profiles/{RND}_pmc_dilnet.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
