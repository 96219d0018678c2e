#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_plan_resweep.log
: > $O
for spec in "AMX_CONV_WS_DGRAD 7,3,5,6" "AMX_WGRAD_WS_MASK 3,7,1" "AMX_CONV_XCD 3,1,0" "AMX_WGRAD_WS_WM4 128,64,256"; do
  set -- $spec
  timeout 600 python tools/gpu_env_ab.py $1 $2 2>&1 | grep "step ms" >> $O
done
cat $O
