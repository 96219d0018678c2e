#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python tools/gpu_step_ab.py "AMX_FIRST_WGRAD_MAIN=1" "AMX_FIRST_WGRAD_MAIN=0" > $O/r06_first_wgrad_main_ab.log 2>&1; echo "ab rc=$?"
tail -3 $O/r06_first_wgrad_main_ab.log
