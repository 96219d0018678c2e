#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 700 python tools/gpu_env_ab.py AMX_BWD_SUMS 3,1,0,2 > $O/r06_bwd_sums_ab2.log 2>&1; echo "ab rc=$?"
timeout 1500 python -m pytest tests/test_seg_gpu.py -q -x -k "loaders or determinism or net_fwd or trajectory or full_width or kernel_level" > $O/r06_bsum_tests2.log 2>&1; echo "tests rc=$?"
tail -4 $O/r06_bwd_sums_ab2.log; tail -4 $O/r06_bsum_tests2.log
