#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -E "^\{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('run', $i, d['value'], d['ms_per_step'], d['allocator'], [x for x in d['step_ms_all'] if x > 25])"
done
