#!/bin/bash
# Round-2 GPU trip U: fused classification head + in-kernel input normalisation: gpu tests, dilnet frame A/B, predict fps
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r02u_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "AMX_FUSE_HEAD=0" "AMX_FUSE_HEAD=1" ) > gpurun_out/r02u_step_ab.log 2>&1
( timeout 300 python tools/gpu_predict_host.py 2>&1 | grep "frames:" ) > gpurun_out/r02u_predict.log 2>&1
echo "== pytest"; tail -3 gpurun_out/r02u_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02u_step_ab.log | tail -3; echo "== predict"; cat gpurun_out/r02u_predict.log
