#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python tools/gpu_step_ab.py "AMX_DGRAD_SPLIT=1" "AMX_DGRAD_SPLIT=0" > $O/r06_dgrad_split_ab.log 2>&1; echo "ab rc=$?"
timeout 1500 python -m pytest tests/test_seg_gpu.py -q -x -k "two_source or loaders or determinism or net_fwd or trajectory or full_width_vs" > $O/r06_split_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r06_dgrad_split_ab.log; tail -3 $O/r06_split_tests.log
