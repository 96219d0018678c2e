#!/bin/bash
# rocprofv3 kernel trace of the config-4 rVAE training step (tools/bench_extra.py rvae, without the A/B leg)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && AMX_RVAE_NO_AB=1 timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_rvae -o rv -- python /root/repo/tools/bench_extra.py rvae ) > gpurun_out/prof_rvae.log 2>&1
python - <<'PY' | tee gpurun_out/prof_rvae_summary.txt
import sqlite3, glob
db = glob.glob('/root/repo/gpurun_out/prof_rvae/**/*results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
steps = 13
for n, cl, td, av, pc in rows[:40]:
    print(f"{n.split('(')[0][:70]:70s} calls {cl:5d} us/step {td/steps/1e3:8.1f} avg {av/1e3:8.1f} {pc:5.1f}%")
print("sum ms/step", sum(r[2] for r in rows) / steps / 1e6)
PY
