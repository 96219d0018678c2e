#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/pmc_$c -o pmc --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
cd /root/repo; ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE; tail -2 gpurun_out/pmc_FETCH_SIZE.log
