#!/bin/bash
# Round-3 trip B: new gpu tests (saved-activation rVAE backward, DilatedBlock dropout, IoU, DKL extractor golden) + rVAE A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
grep -E "MemTotal|MemAvailable" /proc/meminfo > gpurun_out/r03b_meminfo.txt; nproc >> gpurun_out/r03b_meminfo.txt
( timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_gp_gpu.py tests/test_seg_gpu.py -m gpu -q -x -k "rdecoder or config4 or elbo or extractor or dropout or batchnorm or iou or accuracy" 2>&1 | tail -15 ) > gpurun_out/r03b_pytest.log 2>&1
( timeout 600 python tools/bench_extra.py rvae ) > gpurun_out/r03b_rvae.log 2>&1
( AMX_RDEC_FWD_MT=128 timeout 600 python tools/bench_extra.py rvae ) > gpurun_out/r03b_rvae_fwd128.log 2>&1
echo "== pytest"; tail -6 gpurun_out/r03b_pytest.log; echo "== rvae"; tail -1 gpurun_out/r03b_rvae.log; echo "== rvae fwd128"; tail -1 gpurun_out/r03b_rvae_fwd128.log; cat gpurun_out/r03b_meminfo.txt
