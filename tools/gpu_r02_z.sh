#!/bin/bash
# Round-2 GPU trip Z: conv1_wgrad with its accumulators in registers: tests + in-step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r02z_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "lib=prec1" "" ) > gpurun_out/r02z_step_ab.log 2>&1
echo "== pytest"; tail -3 gpurun_out/r02z_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02z_step_ab.log | tail -3
