"""Summarises the PMC pass of tools/gpu_pmc_lds.sh (gpurun_out/pmc_lds) into profiles/<round>_lds_conflicts.md.
   python tools/summarize_pmc_lds.py r06 <git head>"""
import collections, csv, glob, os, re, sys
rnd, head = (sys.argv[1:] + ["r06", "unknown"])[:2]
path = glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    d = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], collections.Counter(), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    d[1][r["Counter_Name"]] += float(r["Counter_Value"])
out = [f"# LDS bank conflicts of the conv kernels' address patterns in isolation (tools/micro/lds_b128_probe.hip), MI355X, round {rnd[1:]}, tree {head}", "",
       "One `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace` pass;",
       "256 workgroups x 4096 repetitions of the pattern per kernel.  `conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (the ratio",
       "`profiles/r05_pmc_sq.md` reports per product kernel); `cycles / instr` = SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS.", "",
       "| pattern | us | LDS instrs | IDX_ACTIVE | BANK_CONFLICT | ADDR_CONFLICT | conflict | cycles / instr |", "|---|---|---|---|---|---|---|---|"]
for name, c, ns in per.values():
    nm = re.sub(r"\(.*", "", name).replace("void ", "")
    ia, bc = c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"]
    out.append(f"| `{nm}` | {ns / 1e3:.0f} | {c['SQ_INSTS_LDS']:.3g} | {ia:.3g} | {bc:.3g} | {c['SQ_LDS_ADDR_CONFLICT']:.3g} | {bc / max(ia, 1):.3f} | {ia / max(c['SQ_INSTS_LDS'], 1):.2f} |")
open(f"profiles/{rnd}_lds_conflicts.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
