#!/bin/bash
# Locator on the MI355X: parity tests, the bench line and a per-kernel rocprofv3 summary (tools/grun.sh 900 'bash tools/gpu_locator_prof.sh <tag>')
cd /root/repo; export TMPDIR=/tmp; TAG=${1:-r03_locator}
(timeout 600 python -m pytest tests/test_locator_gpu.py -q 2>&1 | tail -4) > gpurun_out/${TAG}_pytest.log 2>&1
(timeout 300 python tools/bench_extra.py locate) > gpurun_out/${TAG}_bench.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/${TAG}_prof -o loc -- python /root/repo/tools/bench_extra.py locate) > gpurun_out/${TAG}_rocprof.log 2>&1
cat gpurun_out/${TAG}_pytest.log; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-700
python - <<PY | tee gpurun_out/${TAG}_kernels.txt
import sqlite3
c = sqlite3.connect("gpurun_out/${TAG}_prof/loc_results.db")
print("kernel | calls | avg us")
for r in c.execute("select name,total_calls,total_duration,average from top_kernels").fetchall()[:12]:
    print(r[0].split("(")[0][:48], "|", r[1], "|", round(r[3] / 1e3, 2))
PY
