#!/bin/bash
# Round-5 first GPU trip: the gpu test tier on the tree with the frozen switch table / new kernels, the measured floors of
# the upgraded full-size parity tests, a bench line as this box's baseline, and a coarse plan sweep of the U-Net step.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/r05_pytest_gpu_trip1.log 2>&1
( timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_seg_gpu.py -m gpu -q -s -k "config4_vs_oracle or kernel_level_large or full_width_vs_oracle" 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r05_fullsize_parity_probe.log 2>&1
( time timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r05_bench_trip1.log 2>&1
( timeout 600 python tools/gpu_step_ab.py "" "AMX_CONV_NT=4" "AMX_CONV_NT=4,AMX_CONV_TH=16" "AMX_CONV_TH=16" "AMX_CONV_XCD=1" "AMX_CONV_XCD=0" "AMX_WGRAD_WS_MASK=7" ) > gpurun_out/r05_step_ab_plan.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r05_pytest_gpu_trip1.log; echo "== probe"; tail -12 gpurun_out/r05_fullsize_parity_probe.log | cut -c1-300
echo "== bench"; tail -4 gpurun_out/r05_bench_trip1.log | cut -c1-1200; echo "== ab"; tail -9 gpurun_out/r05_step_ab_plan.log | cut -c1-260
