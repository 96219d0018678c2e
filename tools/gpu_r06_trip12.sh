#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/gpu_step_ab.py "AMX_CONV_XCD=3" "AMX_CONV_XCD=1" > gpurun_out/r06_xcd_default_ab.log 2>&1
tail -2 gpurun_out/r06_xcd_default_ab.log
