#!/bin/bash
# round 6, trip 19: kernel statistics (serial schedule) of the step with the fused pooling backward
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r06_prof_serial -o r06 -- python /root/repo/bench.py --steps 5 --warmup 2 --serial --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 ) > gpurun_out/r06_rocprof_serial.log 2>&1
python - <<'P'
import sqlite3
c = sqlite3.connect("gpurun_out/r06_prof_serial/r06_results.db")
for n, cl, td, av, pc in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:40]:
    if any(k in n for k in ("pool", "conv1", "px_ce", "bn_bwd", "upsample")):
        print(f"{n[:60]:60s} {cl:4d} {av/1e3:9.1f} us")
P
