#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python tools/gpu_step_ab.py "AMX_REDUCE_STREAM=0" "AMX_REDUCE_STREAM=1" > $O/r06_reduce_stream_ab.log 2>&1; echo "ab rc=$?"
tail -3 $O/r06_reduce_stream_ab.log
