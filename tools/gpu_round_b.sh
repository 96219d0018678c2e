#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 900 python tools/bench_extra.py rvae predict ) > gpurun_out/bench_extra.log 2>&1
echo "== pytest"; tail -12 gpurun_out/pytest_gpu.log; echo "== extra"; grep -E "^\{|Error|error" gpurun_out/bench_extra.log | tail -5
