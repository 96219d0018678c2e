#!/bin/bash
# Round-2 GPU trip C: MFMA+LDS micro-benchmark, per-shape A/B of compile-time conv variants, step-level A/B.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python tools/micro/run_mfma_lds.py ) > gpurun_out/r02c_micro.log 2>&1
( timeout 900 python tools/gpu_probe_r02.py ) > gpurun_out/r02c_probe.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "" "lib=wg0" "lib=swp" "lib=exact" "lib=swpexact" ) > gpurun_out/r02c_step_ab.log 2>&1
echo "== micro"; cat gpurun_out/r02c_micro.log | tail -8; echo "== probe"; cat gpurun_out/r02c_probe.log | grep -v Warning | tail -50; echo "== step"; grep -v Warn gpurun_out/r02c_step_ab.log | tail -8
