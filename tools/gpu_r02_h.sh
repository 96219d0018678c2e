#!/bin/bash
# Round-2 GPU trip H: transposed-tail K chunk + 4 waves/SIMD on the TAIL classes (dilnet): tests, in-process A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r02h_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "lib=pretail" "" "AMX_CONV_LATTICE=0" ) > gpurun_out/r02h_step_ab.log 2>&1
( timeout 600 python tools/bench_extra.py segfamily predict ) > gpurun_out/r02h_extra.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r02h_pytest_gpu.log; echo "== step"; grep -v Warn gpurun_out/r02h_step_ab.log | tail -4; echo "== extra"; grep '^{' gpurun_out/r02h_extra.log
