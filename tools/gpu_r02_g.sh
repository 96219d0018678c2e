#!/bin/bash
# Round-2 GPU trip G: lattice-mode dilated convolutions: gpu tests, step-level A/B (AMX_CONV_LATTICE), dilnet kernel trace
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r02g_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "AMX_CONV_LATTICE=0" "AMX_CONV_LATTICE=1" ) > gpurun_out/r02g_step_ab.log 2>&1
( timeout 600 python tools/bench_extra.py segfamily ) > gpurun_out/r02g_segfamily.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02g_prof_predict -o predict -- python /root/repo/tools/bench_extra.py predict ) > gpurun_out/r02g_rocprof_predict.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r02g_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02g_step_ab.log | tail -3; echo "== segfamily"; tail -5 gpurun_out/r02g_segfamily.log
find gpurun_out/r02g_prof_predict -name "*kernel_stats.csv" | head -1 | xargs -r head -25
