#!/bin/bash
# Round-2 GPU trip Q: predict pipeline with high-priority copy streams: throughput A/B + timeline
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for pr in 0 -1; do
  echo "== AMX_COPY_STREAM_PRIORITY=$pr"
  AMX_COPY_STREAM_PRIORITY=$pr timeout 300 python tools/gpu_predict_host.py 2>&1 | grep "frames:"
done 2>&1 | tee gpurun_out/r02q_predict_prio.log
bash tools/gpu_r02_p.sh 2>&1 | tail -12
