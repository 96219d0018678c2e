"""Turns the scratch outputs of tools/gpu_r02_final.sh (gpurun_out/) into the committed summaries of a round:
   profiles/rNN_unet_bs32_512_kernel_stats.md / ..._serial.md   (rocprofv3 --kernel-trace --stats)
   profiles/rNN_pmc_hbm_traffic.json                            (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
   usage: python tools/summarize_profiles.py r02 <git head of the measured tree>
"""
import collections, csv, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")


def kernel_stats(steps=7, sub="prof", outname="r01_unet_bs32_512_kernel_stats_final.md", note="", db="r01_results.db", rnd="1", head=""):
    c = sqlite3.connect(os.path.join(G, sub, db))
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 (7 training steps), MI355X, round " + rnd + (f", tree {head}" if head else "") + "\n\n", note,
           "U-Net nb_classes=3, 512x512, bs=32, fp32.  " + ("" if note else "Weight-gradient kernels run on a second stream, "
           "so kernel times overlap (and are longer than stand-alone) and their sum exceeds the step time; the "
           "serialised companion is the `..._serial.md` file of the same round.") + "\n\n",
           "| kernel | calls | total us | avg us | % | us/step |\n|---|---|---|---|---|---|\n"]
    for n, cl, td, av, pc in rows:
        out.append(f"| `{short(n)[:72]}` | {cl} | {td:.0f} | {av:.1f} | {pc:.2f} | {td / steps:.0f} |\n")
    out.append(f"\nSum of kernel time {tot / 1e3 / steps:.2f} ms/step.\n")
    # busy/idle of a steady-state step
    ks = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
    ad = [i for i, k in enumerate(ks) if k[0].startswith("adam_flat")]
    if len(ad) >= 5:
        R = ks[ad[3] + 1:ad[4] + 1]
        s0, e1 = min(r[1] for r in R), max(r[2] for r in R)
        iv = sorted((r[1], r[2]) for r in R)
        busy, cs, ce = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > ce:
                busy += ce - cs; cs, ce = a, b
            else:
                ce = max(ce, b)
        busy += ce - cs
        main = sum(r[2] - r[1] for r in R if r[3] == 0) / 1e6
        side = sum(r[2] - r[1] for r in R if r[3] != 0) / 1e6
        out.append(f"\nSteady-state step (5th of the trace): wall {(e1 - s0) / 1e6:.2f} ms, GPU busy (union of kernel intervals) "
                   f"{busy / 1e6:.2f} ms, main-stream kernel time {main:.2f} ms, side-stream (weight gradients) {side:.2f} ms.\n")
    fam = [(cl, td) for n, cl, td, av, pc in rows if short(n).startswith(("conv_fwd_kernel", "conv_ws_kernel"))]
    if fam:
        out.append(f"\nconv_fwd_kernel + conv_ws_kernel family (amx_conv2d_fwd + amx_conv2d_dgrad): {sum(c_ for c_, _ in fam)} launches, "
                   f"average {sum(t for _, t in fam) / sum(c_ for c_, _ in fam) / 1e3:.4f} ms per launch.\n")
    open(os.path.join(ROOT, "profiles", outname), "w").writelines(out)


def pmc_traffic(tag="r01", head="", sub_prefix="pmc_"):
    def load(which):
        d = collections.defaultdict(lambda: [0, 0.0])
        order = []
        for r in csv.DictReader(open(os.path.join(G, f"{sub_prefix}{which}", "pmc_counter_collection.csv"))):
            if r["Counter_Name"] != which:
                continue
            order.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
        order.sort()
        names = [o[1] for o in order]
        ad = [i for i, n in enumerate(names) if n.startswith("adam_flat")]
        sel = order[ad[0] + 1:ad[1] + 1] if len(ad) > 1 else order      # the second (steady) step
        for _, n, v in sel:
            d[short(n)][0] += 1
            d[short(n)][1] += v
        return d
    f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
    kern = {}
    for n in f:
        cnt = f[n][0]
        kern[n] = {"launches_per_step": cnt, "fetch_MB_per_launch": round(2 * f[n][1] * 1024 / 1e6 / cnt, 1),
                   "write_MB_per_launch": round(w[n][1] * 1024 / 1e6 / max(w[n][0], 1), 1)}
    fam = [k for k in kern if k.startswith(("conv_fwd_kernel", "conv_ws_kernel"))]
    L = sum(kern[k]["launches_per_step"] for k in fam)
    fe = sum(kern[k]["fetch_MB_per_launch"] * kern[k]["launches_per_step"] for k in fam)
    wr = sum(kern[k]["write_MB_per_launch"] * kern[k]["launches_per_step"] for k in fam)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                     "--steps 1 --warmup 1 (second = steady training step), MI355X, round " + tag[1:].lstrip("0"),
           "git_head": head,
           "corrections": "FETCH_SIZE (KB) doubled: on gfx950 it reports 1/2 of a wide coalesced stream "
                          "(MI355X_MICROARCH.md HBM section); calibrated on pool_fwd_kernel (algorithmic 313 MB read / 78 MB "
                          "written per top-level launch).  WRITE_SIZE (KB) used as is.",
           "kernels": kern,
           "conv_fwd_family": {"launches_per_step": L, "hbm_MB_per_launch": round((fe + wr) / L, 1),
                               "fetch_GB_per_step": round(fe / 1e3, 2), "write_GB_per_step": round(wr / 1e3, 2),
                               "algorithmic_GB_per_step": {"read": 5.41, "write": 5.41,
                                                           "note": "each conv reads its input once and writes its output "
                                                                   "once (fwd 3.06+2.35, dgrad 2.35+3.06)"}},
           "step_total": {"fetch_GB": round(sum(v["fetch_MB_per_launch"] * v["launches_per_step"] for v in kern.values()) / 1e3, 2),
                          "write_GB": round(sum(v["write_MB_per_launch"] * v["launches_per_step"] for v in kern.values()) / 1e3, 2)}}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out["conv_fwd_family"]), out["step_total"])
    for k in ("pool_fwd_kernel", "bn_bwd_apply_kernel", "upsample_bwd_kernel", "pool_bwd_kernel"):
        print(k, kern.get(k))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    head = sys.argv[2] if len(sys.argv) > 2 else ""
    rnd = tag[1:].lstrip("0")
    pre = "" if tag == "r01" else tag + "_"
    db = f"{tag}_results.db"
    if os.path.exists(os.path.join(G, pre + "prof", db)):
        kernel_stats(sub=pre + "prof", outname=f"{tag}_unet_bs32_512_kernel_stats.md", db=db, rnd=rnd, head=head)
    if os.path.exists(os.path.join(G, pre + "prof_serial", db)):
        kernel_stats(sub=pre + "prof_serial", outname=f"{tag}_unet_bs32_512_kernel_stats_serial.md", db=db, rnd=rnd, head=head,
                     note="**`--serial`: every kernel on ONE stream** — the mode of bench.py's per-kernel HIP-event pass "
                          "(`roofline.avg_launch_ms`), which this summary must agree with.\n\n")
    if os.path.exists(os.path.join(G, pre + "pmc_FETCH_SIZE")):
        pmc_traffic(tag, head, pre + "pmc_")
