#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_reference_suite_gpu.py -x -q > gpurun_out/r06_ski_pytest.log 2>&1
tail -3 gpurun_out/r06_ski_pytest.log
timeout 600 python tools/gpu_lu_probe.py 2500 > gpurun_out/r06_lu_probe.log 2>&1
cat gpurun_out/r06_lu_probe.log | tail -60
