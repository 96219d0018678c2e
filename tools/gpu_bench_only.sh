#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_quick.log 2>&1
( timeout 600 python tools/bench_extra.py rvae ) > gpurun_out/bench_rvae_quick.log 2>&1
grep -E "^\{" gpurun_out/bench_quick.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('UNET', d['value'], d['ms_per_step'], d['step_frac_of_mfma_f32_peak'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline_wgrad']['frac'], d['roofline_wgrad']['ms_per_step'])"
grep -E "^\{" gpurun_out/bench_rvae_quick.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('RVAE', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms'], d['roofline_fwd']['frac'], d['roofline_fwd']['ms'])"
tail -3 gpurun_out/bench_quick.log | grep -iE "error|Traceback" 
