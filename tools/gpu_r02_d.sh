#!/bin/bash
# Round-2 GPU trip D: gpu tests of the new features, MFMA ceiling with random operands, conv occupancy sweep, DKL builder A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r02d_pytest_gpu.log 2>&1
( timeout 300 python tools/micro/run_mfma_lds.py ) > gpurun_out/r02d_micro.log 2>&1
( timeout 600 python tools/gpu_probe_r02.py occ ) > gpurun_out/r02d_probe_occ.log 2>&1
( timeout 300 python tools/gpu_km_ab.py ) > gpurun_out/r02d_km_ab.log 2>&1
echo "== pytest"; tail -14 gpurun_out/r02d_pytest_gpu.log; echo "== micro"; tail -10 gpurun_out/r02d_micro.log; echo "== occ"; grep -v Warn gpurun_out/r02d_probe_occ.log | tail -30; echo "== km"; tail -8 gpurun_out/r02d_km_ab.log
