#!/bin/bash
# Round-2 GPU trip E: per-phase wave timing of the conv kernel, MFMA ceiling with random operands, dilated 8-row tiles
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python tools/gpu_conv_phases.py ) > gpurun_out/r02e_phases.log 2>&1
( timeout 300 python tools/micro/run_mfma_lds.py ) > gpurun_out/r02e_micro.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "" "AMX_CONV_DIL_TH=8" ) > gpurun_out/r02e_step_ab.log 2>&1
echo "== phases"; grep -v Warn gpurun_out/r02e_phases.log | tail -80; echo "== micro"; tail -5 gpurun_out/r02e_micro.log; echo "== step"; grep -v Warn gpurun_out/r02e_step_ab.log | tail -4
