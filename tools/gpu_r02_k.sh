#!/bin/bash
# Round-2 GPU trip K: wgrad row-sweep unrolling (1 / 2 / 4 rows) vs the rolled loop, in-step A/B + phases of the default
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python tools/gpu_step_ab.py "lib=prewg" "lib=u1" "" "lib=u4" ) > gpurun_out/r02k_step_ab.log 2>&1
( timeout 600 python tools/gpu_wgrad_phases.py ) > gpurun_out/r02k_wgrad_phases.log 2>&1
echo "== step"; grep -v Warn gpurun_out/r02k_step_ab.log | tail -5; echo "== phases"; grep "==\|MFMA\|issue" gpurun_out/r02k_wgrad_phases.log
