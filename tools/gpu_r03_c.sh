#!/bin/bash
# Round-3 trip C: 48 + 4 column class A/B on dilnet, dilated tests, the full 4096-frame config-3 run
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -k "dilated or dilnet or head or dsum or predictor or with_dilation" 2>&1 | tail -8 ) > gpurun_out/r03c_pytest.log 2>&1
( timeout 600 python tools/gpu_nt3_ab.py nt3w2 ) > gpurun_out/r03c_nt3_ab.log 2>&1
( timeout 900 python tools/bench_extra.py predict4096 ) > gpurun_out/r03c_predict4096.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r03c_pytest.log; echo "== nt3"; cat gpurun_out/r03c_nt3_ab.log | grep lib=; echo "== 4096"; tail -2 gpurun_out/r03c_predict4096.log
