#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes, kernel trace only) of the dominant kernels of the other
# workloads: rVAE decoder kernels, DKL covariance builder, Locator passes, dilnet predict (one 16-frame chunk).
# Summarised into profiles/<round>_pmc_hbm_extra.md (corrections as in tools/summarize_profiles.py: FETCH_SIZE doubled).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; RND=${1:-r03}
cat > /tmp/dil16.py <<'PY'
import sys; sys.path.insert(0, '/root/repo')
import torch, atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1); net = net.cuda().eval()
x = torch.rand(16, 1, 1024, 1024, device="cuda")
for _ in range(2): predict_proba(net, x)
torch.cuda.synchronize()
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  AMX_RVAE_NO_AB=1 timeout 600 rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/pmcx_$c -o pmc --output-format csv -- python /root/repo/tools/bench_extra.py rvae dkl locate > /root/repo/gpurun_out/pmcx_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/pmcx_dil_$c -o pmc --output-format csv -- python /tmp/dil16.py > /root/repo/gpurun_out/pmcx_dil_$c.log 2>&1
done
cd /root/repo; python tools/summarize_pmc_hbm_extra.py $RND
