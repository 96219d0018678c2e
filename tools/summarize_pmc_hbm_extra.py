"""Summary of tools/gpu_pmc_hbm_extra.sh -> profiles/<round>_pmc_hbm_extra.md: HBM bytes per launch of the dominant kernels of
the rVAE / DKL / Locator / dilnet-predict workloads next to their algorithmic bytes."""
import collections, csv, glob, re, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"


def load(sub, which):
    path = glob.glob(f"gpurun_out/{sub}/**/*counter_collection.csv", recursive=True)
    d = collections.OrderedDict()
    if not path:
        return d
    for r in csv.DictReader(open(path[0])):
        if r["Counter_Name"] != which:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        d.setdefault(name, []).append(float(r["Counter_Value"]))
    return d


def mb(vals, fetch):           # KB counters; FETCH_SIZE reports half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md)
    v = vals[-3:] if len(vals) >= 3 else vals
    return (2.0 if fetch else 1.0) * sum(v) / len(v) * 1024 / 1e6


rows = []
ALG = {"rdecoder_fwd_kernel<128, 64>": "reads coordinates / z (< 1 MB), writes x_rec 8 MB + the two saved hidden images 2 x 1074 MB",
       "rdecoder_bwd_kernel<128, 64, 2, true>": "reads the saved hidden images 2 x 1074 MB + d x_rec 8 MB, writes per-sample partial rows (~70 MB)",
       "kernel_matrix_kernel<float>": "writes K: 16384^2 x 4 B = 1074 MB",
       "locate_tile_kernel": "reads 32 x 1024^2 probabilities = 134 MB, writes ~0.7 B / pixel",
       "conv1_fwd_kernel<true, true>": "reads 16 x 1024^2 x 4 B = 67 MB, writes c1 1879 MB + pooled 470 MB",
       "upsample_fwd_kernel": "reads 470 MB, writes 1879 MB"}
for sub in ("pmcx", "pmcx_dil"):
    f, w = load(f"{sub}_FETCH_SIZE", "FETCH_SIZE"), load(f"{sub}_WRITE_SIZE", "WRITE_SIZE")
    for k in f:
        if not any(t in k for t in ("rdecoder", "kernel_matrix_kernel", "locate_", "conv_fwd_kernel", "conv1_fwd", "upsample_fwd")):
            continue
        rows.append((k, len(f[k]), mb(f[k], True), mb(w.get(k, [0.0]), False), ALG.get(k, "")))
out = [f"# HBM traffic of the dominant kernels of the other workloads, MI355X, {RND}", "",
       "`tools/gpu_pmc_hbm_extra.sh`: `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only) over "
       "`tools/bench_extra.py rvae dkl locate` and one 16-frame dilnet chunk; averages of the last three launches of a kernel; "
       "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, as calibrated in `r0x_pmc_hbm_traffic.json`).", "",
       "| kernel | launches seen | fetched MB / launch | written MB / launch | algorithmic |", "|---|---|---|---|---|"]
for k, n, fm, wm, alg in rows:
    out.append(f"| `{k[:60]}` | {n} | {fm:.0f} | {wm:.0f} | {alg} |")
out += ["", "Caveat: the doubling of FETCH_SIZE was calibrated on wide sequential streams (pool_fwd: 313 MB algorithmic).  The 1x1 "
        "convolution above streams its 872 MB input once and reads 1440 MB by this accounting, so for narrower access patterns "
        "the factor may over-count by up to 2x; the WRITE_SIZE column and the RATIOS between kernels of one pattern are reliable "
        "(dilation 6 fetches 2.4x what dilations 2 / 4 fetch for the same tensor)."]
open(f"profiles/{RND}_pmc_hbm_extra.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
