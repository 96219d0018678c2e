#!/bin/bash
# How sensitive is the (Python-launched) step to a busy host?  Runs bench with and without CPU hogs on every core.
cd /root/repo; export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -E "^\{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['step_ms_median_max'])"; }
run quiet
N=$(nproc)
pids=()
for i in $(seq 1 $N); do ( timeout 60 python3 -c "
while True: pass" ) & pids+=($!); done
sleep 3
run hogs_x1
for p in "${pids[@]}"; do kill $p 2>/dev/null; done
wait 2>/dev/null
echo nproc $N
