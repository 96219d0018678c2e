#!/bin/bash
# Round-3 trip A: the RCCL branch on silicon first, then the whole gpu tier and a baseline bench of the unchanged kernels.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_nccl_gpu.py -m gpu -q -x 2>&1 | tail -40 ) > gpurun_out/r03a_nccl.log 2>&1
( AMX_BENCH_FORCE_DP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --no-extra --no-cpu-baseline --sustain-seconds 0 ) > gpurun_out/r03a_bench_forcedp.log 2>&1
( timeout 600 python bench.py --no-extra --no-cpu-baseline --sustain-seconds 3 ) > gpurun_out/r03a_bench.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r03a_pytest_gpu.log 2>&1
echo "== nccl"; tail -15 gpurun_out/r03a_nccl.log; echo "== forcedp"; tail -2 gpurun_out/r03a_bench_forcedp.log | cut -c1-900; echo "== bench"; tail -1 gpurun_out/r03a_bench.log | cut -c1-600; echo "== pytest"; tail -5 gpurun_out/r03a_pytest_gpu.log
