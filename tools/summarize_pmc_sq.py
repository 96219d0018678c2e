"""Summarises the two SQ PMC passes of tools/gpu_pmc_sq.sh (gpurun_out/pmc_sq, pmc_sq2) per kernel of the SECOND
(steady) training step into profiles/<round>_pmc_sq.md.   python tools/summarize_pmc_sq.py r02 <git head>"""
import collections, csv, glob, os, re, sys
rnd, head = (sys.argv[1:] + ["r02", "unknown"])[:2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(sub):
    path = glob.glob(os.path.join(ROOT, "gpurun_out", sub, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.OrderedDict()                 # dispatch -> (kernel, {counter: value}, duration)
    for r in csv.DictReader(open(path)):
        d = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
        d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(per.values())


def steady(disp):
    """dispatches of the last training step: from the last conv1_fwd_kernel on."""
    idx = [i for i, d in enumerate(disp) if d[0].startswith("conv1_fwd_kernel")]
    return disp[idx[-1]:] if idx else disp


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:60]


a, b = steady(load("pmc_sq")), steady(load("pmc_sq2"))
agg = collections.OrderedDict()
for src in (a, b):
    for name, ctr, dur in src:
        e = agg.setdefault(short(name), collections.Counter())
        for k, v in ctr.items():
            e[k] += v
        e["_n_" + ("a" if src is a else "b")] += 1
        e["_ns_" + ("a" if src is a else "b")] += dur
rows = []
for name, e in agg.items():
    if not e.get("SQ_BUSY_CYCLES"):
        continue
    wave = e["SQ_WAVE_CYCLES"] or 1
    rows.append((e["_ns_a"], name, e["_n_a"],
                 e["SQ_VALU_MFMA_BUSY_CYCLES"] / max(e.get("GRBM_GUI_ACTIVE", 0) * 128.0, 1.0),   # per SIMD-cycle: GUI_ACTIVE is summed
                 #                                   over the 8 XCDs; 1024 SIMDs -> SIMD-cycles = GUI_ACTIVE / 8 * 1024
                 e["SQ_ACTIVE_INST_ANY"] / wave, e["SQ_WAIT_INST_ANY"] / wave, e["SQ_WAIT_INST_LDS"] / wave,
                 e["SQ_LDS_BANK_CONFLICT"] / max(e.get("SQ_LDS_IDX_ACTIVE", 0), 1),
                 e.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), e.get("GRBM_GUI_ACTIVE", 0)))
rows.sort(reverse=True)
out = [f"# SQ counters of the U-Net training step (bs 32, 512², serial schedule), MI355X, round {rnd[1:]}, tree {head}", "",
       "`tools/gpu_pmc_sq.sh`: two `rocprofv3 --pmc ... --kernel-trace` passes (no other trace domain), counters summed over the",
       "dispatches of one steady training step per kernel.  `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);",
       "`issuing` / `waiting` / `waiting on LDS` = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_INST_LDS per SQ_WAVE_CYCLES;",
       "`LDS conflicts` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; `MFMA MOPS` = SQ_INSTS_VALU_MFMA_MOPS_F32 (512 FLOP each).", "",
       "| kernel | launches | ms | MFMA busy | issuing | waiting | waiting on LDS | LDS conflicts | MFMA MOPS | TFLOP/s from MOPS |", "|---|---|---|---|---|---|---|---|---|---|"]
for ns, name, n, mf, act, wt, wl, bc, mops, gui in rows[:24]:
    tf = mops * 512 / (ns * 1e-9) / 1e12 if ns else 0
    out.append(f"| `{name}` | {n} | {ns/1e6:.3f} | {mf:.3f} | {act:.3f} | {wt:.3f} | {wl:.3f} | {bc:.3f} | {mops:.3g} | {tf:.1f} |")
open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_sq.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
