#!/bin/bash
cd /root/repo
for h in kernel memcpy; do for d in kernel memcpy; do
  echo "== H2D=$h D2H=$d"; AMX_PRED_H2D=$h AMX_PRED_D2H=$d timeout 300 python tools/gpu_predict_check.py 2>&1 | grep -E "rep|Assert|coords ok" | tr '\n' ' '; echo
done; done
