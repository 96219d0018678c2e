#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python tools/gpu_env_ab.py AMX_BWD_SUMS 1,3,5,7,0 > $O/r06_bwd_sums_ab3.log 2>&1; echo "ab rc=$?"
tail -5 $O/r06_bwd_sums_ab3.log
