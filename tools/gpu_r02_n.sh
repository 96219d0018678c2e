#!/bin/bash
# Round-2 GPU trip N: cheaper HBM-bound passes (stats merge, head kernels, first-layer conv): tests + in-step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r02n_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "lib=premisc" "" ) > gpurun_out/r02n_step_ab.log 2>&1
echo "== pytest"; tail -3 gpurun_out/r02n_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02n_step_ab.log | tail -3
