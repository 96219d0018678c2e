#!/bin/bash
# Round-2 GPU trip Y: weight image by LDS-DMA (global_load_lds) with 4 / 5 waves per SIMD: per-layer and in-step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python tools/gpu_probe_r02.py unet dilnet ) > gpurun_out/r02y_probe.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "" "lib=glds4" "lib=glds5" ) > gpurun_out/r02y_step_ab.log 2>&1
echo "== probe"; grep -v Warn gpurun_out/r02y_probe.log | tail -34; echo "== step"; grep -v Warn gpurun_out/r02y_step_ab.log | tail -4
