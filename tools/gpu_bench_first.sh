#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -E "^\{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('first', d['value'], d['ms_per_step'], d['step_ms_all'])"
