#!/bin/bash
# new-feature validation: locator / checkpoints / conv-encoder tests + locator bench
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_locator_gpu.py tests/test_checkpoints_gpu.py tests/test_vae_gpu.py -m gpu -x -q ) > gpurun_out/pytest_new.log 2>&1
tail -5 gpurun_out/pytest_new.log
( timeout 600 python tools/bench_extra.py locate ) > gpurun_out/bench_locate.log 2>&1
tail -3 gpurun_out/bench_locate.log | cut -c1-1500
