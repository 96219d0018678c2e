#!/bin/bash
# round-6 first measurement trip: CU-mask probe + step A/B, new parity tests, config-5 fit step, LDS probe
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 120 tools/micro/cumask_probe.bin > $O/r06_cumask_probe.log 2>&1; echo "probe rc=$?"
timeout 600 python tools/gpu_cumask_ab.py > $O/r06_cumask_ab.log 2>&1; echo "cumask_ab rc=$?"
timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_vae_gpu.py tests/test_gp_gpu.py -q -s -k "kink_free or linear_exact or tanh or gp" > $O/r06_new_tests.log 2>&1; echo "tests rc=$?"
timeout 600 python tools/bench_extra.py dklfit > $O/r06_dklfit.log 2>&1; echo "dklfit rc=$?"
timeout 400 tools/gpu_pmc_lds.sh r06 $1 > $O/r06_pmc_lds.log 2>&1; echo "lds rc=$?"
tail -5 $O/r06_cumask_probe.log; tail -14 $O/r06_cumask_ab.log; tail -12 $O/r06_new_tests.log; tail -3 $O/r06_dklfit.log | cut -c1-1500
