#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python tools/gpu_rvae_ab.py rdprio rdstag > $O/r06_rvae_asym_ab.log 2>&1; echo "rvae rc=$?"
timeout 600 python tools/bench_extra.py dkl > $O/r06_dkl2.log 2>&1; echo "dkl rc=$?"
timeout 900 python -m pytest tests/test_gp_gpu.py -q > $O/r06_gp_tests.log 2>&1; echo "gp tests rc=$?"
tail -8 $O/r06_rvae_asym_ab.log; tail -1 $O/r06_dkl2.log | cut -c1-900; tail -3 $O/r06_gp_tests.log
