#!/bin/bash
# round 6, trip 16: KISS-GP with the Kronecker low-rank core — DKL gpu tests, config-5 fit step (both GP layers)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/gpu_kron_host_probe.py > gpurun_out/r06_kron_host_probe.log 2>&1
cat gpurun_out/r06_kron_host_probe.log | tail -30
timeout 1500 python -m pytest tests/test_gp_gpu.py tests/test_reference_suite_gpu.py -x -q -k "gp or dkl or kiss or config5" > gpurun_out/r06_ski_pytest.log 2>&1
tail -3 gpurun_out/r06_ski_pytest.log
timeout 900 python tools/bench_extra.py dklfit > gpurun_out/r06_dklfit_ski.log 2>&1
tail -c 1500 gpurun_out/r06_dklfit_ski.log | head -c 1300
