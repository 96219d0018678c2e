#!/bin/bash
# round 6, trip 17: head + loss of the training step in one pass (amx_px_ce_train) — tests, in-process step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_seg_gpu.py -x -q -k "one_pass or net_fwd_bwd_adam or determinism or config1 or config2 or variants" > gpurun_out/r06_pxloss_pytest.log 2>&1
tail -3 gpurun_out/r06_pxloss_pytest.log
timeout 900 python tools/gpu_step_ab.py "AMX_FUSE_PX_LOSS=0" "AMX_FUSE_PX_LOSS=1" > gpurun_out/r06_px_loss_ab.log 2>&1
tail -2 gpurun_out/r06_px_loss_ab.log
