#!/bin/bash
# does an XCD-aware tile order on EVERY convolution launch (AMX_CONV_XCD=1; default 3 = dilated launches
# only) reduce the HBM fetch of the U-Net step's convolution family?  One FETCH_SIZE PMC pass (kernel trace only) of a
# steady step under each setting; the step-time side of the question is profiles/r05_logs/r05_step_ab_plan.log.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for x in 3 1; do
  AMX_CONV_XCD=$x timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/r05_pmc_xcd$x -o pmc --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 > /root/repo/gpurun_out/r05_pmc_xcd$x.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, collections
for x in (3, 1):
    rows = []
    for r in csv.DictReader(open(f"gpurun_out/r05_pmc_xcd{x}/pmc_counter_collection.csv")):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    # the second training step = the dispatches after the first adam_flat_kernel up to the second one
    ad = [i for i, r in enumerate(rows) if r[1].startswith("adam_flat")]
    R = rows[ad[0] + 1:ad[1] + 1] if len(ad) >= 2 else rows
    fam = collections.defaultdict(float)
    for _, k, v in R:
        key = "conv family" if k.startswith(("void conv_fwd_kernel", "void conv_ws_kernel", "conv_fwd_kernel", "conv_ws_kernel")) else ("wgrad family" if "wgrad" in k else "other")
        fam[key] += 2 * v * 1024 / 1e9              # KB -> GB, x2 (gfx950 correction, MI355X_MICROARCH.md)
    print(f"AMX_CONV_XCD={x}: fetch per step, GB: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(fam.items())) + f", total {sum(fam.values()):.2f}")
PY
