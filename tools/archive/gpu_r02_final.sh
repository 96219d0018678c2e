#!/bin/bash
# Round-2 evidence trip: everything the judge reads, from ONE tree (its git head is passed in $1 and stamped on the outputs).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
HEAD=${1:-unknown}
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r02_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02_smoke.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_n1.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof_serial -o r02 -- python /root/repo/bench.py --steps 5 --warmup 2 --serial --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 ) > gpurun_out/r02_rocprof_serial.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof -o r02 -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 ) > gpurun_out/r02_rocprof.log 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/r02_pmc_$c -o pmc --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0 > /root/repo/gpurun_out/r02_pmc_$c.log 2>&1
done
cd /root/repo
( timeout 900 python tools/bench_extra.py predict rvae dkl locate segfamily ) > gpurun_out/r02_bench_extra.log 2>&1
echo $HEAD > gpurun_out/r02_head.txt
echo "== pytest"; tail -3 gpurun_out/r02_pytest_gpu.log; echo "== smoke"; tail -2 gpurun_out/r02_smoke.log; echo "== bench"; tail -5 gpurun_out/r02_bench_n1.log | cut -c1-1500; echo "== rocprof"; ls gpurun_out/r02_prof_serial gpurun_out/r02_prof gpurun_out/r02_pmc_FETCH_SIZE 2>&1 | head -12
