#!/bin/bash
# round-6 second trip: CU-mask A/B off the null stream, staging-map variant libraries, kink-free tests, LDS probe
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 700 python tools/gpu_cumask_ab.py > $O/r06_cumask_ab2.log 2>&1; echo "cumask_ab rc=$?"
timeout 700 python tools/gpu_step_ab.py "" "lib=smap" "lib=wsmap" > $O/r06_stage_map_ab.log 2>&1; echo "stage map rc=$?"
timeout 900 python -m pytest tests/test_seg_gpu.py -q -s -k "kink_free" > $O/r06_kinkfree_tests.log 2>&1; echo "tests rc=$?"
timeout 400 tools/gpu_pmc_lds.sh r06 $1 > $O/r06_pmc_lds.log 2>&1; echo "lds rc=$?"
tail -16 $O/r06_cumask_ab2.log; tail -4 $O/r06_stage_map_ab.log; grep "kink-free\|passed\|failed" $O/r06_kinkfree_tests.log; tail -8 $O/r06_pmc_lds.log
