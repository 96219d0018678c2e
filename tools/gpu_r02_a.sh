#!/bin/bash
# Round-2 GPU trip A: gpu-tier tests, smoke, the full default bench line, serialised rocprof kernel stats.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r02_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02_smoke.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r02_bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof_serial -o r02 -- python /root/repo/bench.py --steps 5 --warmup 2 --serial --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 ) > gpurun_out/r02_rocprof_serial.log 2>&1
echo "== pytest"; tail -15 gpurun_out/r02_pytest_gpu.log; echo "== smoke"; tail -3 gpurun_out/r02_smoke.log; echo "== bench"; tail -8 gpurun_out/r02_bench.log; echo "== rocprof"; tail -3 gpurun_out/r02_rocprof_serial.log; ls gpurun_out/r02_prof_serial 2>/dev/null | head
