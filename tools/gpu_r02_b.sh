#!/bin/bash
# Round-2 GPU trip B: gpu tests on the persistent convolution, per-shape A/B vs the round-1 library, step-level A/B.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r02b_pytest_gpu.log 2>&1
( timeout 600 python tools/gpu_probe_r02.py ) > gpurun_out/r02b_probe.log 2>&1
( timeout 600 python tools/gpu_step_ab.py "AMX_CONV_PERSIST=0" "AMX_CONV_PERSIST=1" "AMX_CONV_PERSIST=2" "AMX_CONV_PERSIST=1,AMX_CONV_WRES=0" "AMX_CONV_PERSIST=0,AMX_CONV_WRES=0" ) > gpurun_out/r02b_step_ab.log 2>&1
echo "== pytest"; tail -12 gpurun_out/r02b_pytest_gpu.log; echo "== probe"; cat gpurun_out/r02b_probe.log | grep -v Warning | tail -50; echo "== step"; grep -v Warn gpurun_out/r02b_step_ab.log | tail -8
