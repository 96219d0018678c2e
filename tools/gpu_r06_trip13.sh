#!/bin/bash
# round 6, trip 13: KISS-GP layer (csrc/ski.hip) — gpu tests of the DKL path, config-5 fit step for both GP layers, headline
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gp_gpu.py tests/test_reference_suite_gpu.py -x -q > gpurun_out/r06_ski_pytest.log 2>&1
tail -5 gpurun_out/r06_ski_pytest.log
timeout 900 python tools/bench_extra.py dklfit > gpurun_out/r06_dklfit_ski.log 2>&1
tail -c 3000 gpurun_out/r06_dklfit_ski.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r06_bench_xcd1.log 2>&1
tail -c 1500 gpurun_out/r06_bench_xcd1.log
