#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 700 python tools/gpu_env_ab.py AMX_BWD_SUMS 1,0 > $O/r06_bwd_sums_ab.log 2>&1; echo "ab rc=$?"
timeout 1500 python -m pytest tests/test_seg_gpu.py -q -x -k "loaders or determinism or net_fwd or trajectory or full_width" > $O/r06_bsum_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r06_bwd_sums_ab.log; tail -4 $O/r06_bsum_tests.log
