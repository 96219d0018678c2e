#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_predict -o pr -- python /root/repo/tools/bench_extra.py predict ) > gpurun_out/prof_predict.log 2>&1
tail -2 gpurun_out/prof_predict.log | cut -c1-400
python - <<'PY'
import sqlite3, glob
db = glob.glob('/root/repo/gpurun_out/prof_predict/*results.db')[0]
c = sqlite3.connect(db)
for n, cl, td, av, pc in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:12]:
    print(f"{n.split('(')[0][:60]:60s} calls {cl:5d} total {td:10.0f} avg {av:8.1f} {pc:5.1f}%")
PY
