#!/bin/bash
# Hardware-counted MFMA work of one dilnet forward (16 frames of 1024^2) vs the algorithmic FLOP count: how much of the
# issued matrix work is padding?  One PMC pass, kernel trace only.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/dil_pmc.py <<'PY'
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, atomai_amd as aoi
from atomai_amd.nets.fcnn import predict_proba
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net = net.cuda().eval()
x = torch.rand(16, 1, 1024, 1024, device="cuda")
for _ in range(2): predict_proba(net, x)
torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /root/repo/gpurun_out/pmc_dilnet -o pmc --output-format csv -- python /tmp/dil_pmc.py > /root/repo/gpurun_out/pmc_dilnet.log 2>&1
cd /root/repo
python tools/summarize_pmc_dilnet.py ${1:-r03}
