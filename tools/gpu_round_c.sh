#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gp_gpu.py -m gpu -q 2>&1 | tail -30 ) > gpurun_out/pytest_gp_gpu.log 2>&1
( timeout 600 python tools/bench_extra.py dkl ) > gpurun_out/bench_dkl.log 2>&1
echo "== pytest"; tail -8 gpurun_out/pytest_gp_gpu.log; echo "== dkl"; grep -E "^\{|Error|error" gpurun_out/bench_dkl.log | tail -5
