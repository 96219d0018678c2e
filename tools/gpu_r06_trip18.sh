#!/bin/bash
# round 6, trip 18: pooling backward fused with the first layer's weight gradient (amx_pool2x2_bwd_wgrad1) — tests, step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_seg_gpu.py -x -q -k "first_layer_weight or one_pass or net_fwd_bwd_adam or determinism or config1 or config2 or variants or full_width" > gpurun_out/r06_poolwg1_pytest.log 2>&1
tail -3 gpurun_out/r06_poolwg1_pytest.log
timeout 900 python tools/gpu_step_ab.py "AMX_FUSE_POOL_WGRAD1=0" "AMX_FUSE_POOL_WGRAD1=1" > gpurun_out/r06_pool_wgrad1_ab.log 2>&1
tail -2 gpurun_out/r06_pool_wgrad1_ab.log
