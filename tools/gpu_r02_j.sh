#!/bin/bash
# Round-2 GPU trip J: wgrad loaders with tile-independent descriptors: phases, in-step A/B, gpu tests of the conv layers
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python tools/gpu_wgrad_phases.py ) > gpurun_out/r02j_wgrad_phases.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "lib=prewg" "" ) > gpurun_out/r02j_step_ab.log 2>&1
( timeout 1500 python -m pytest tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r02j_pytest_gpu.log 2>&1
echo "== pytest"; tail -3 gpurun_out/r02j_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02j_step_ab.log | tail -3; echo "== phases"; grep -v amdgpu.ids gpurun_out/r02j_wgrad_phases.log
