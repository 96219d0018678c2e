#!/bin/bash
# One GPU trip: gpu-tier tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -o r01 -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/rocprof.log 2>&1
echo "== pytest"; tail -15 gpurun_out/pytest_gpu.log; echo "== smoke"; tail -3 gpurun_out/smoke.log; echo "== bench"; tail -3 gpurun_out/bench.log; echo "== rocprof"; tail -3 gpurun_out/rocprof.log; ls gpurun_out/prof 2>/dev/null | head
