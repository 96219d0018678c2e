#!/bin/bash
# HBM fetch of the dilnet forward's convolution launches (16 frames of 1024^2) with / without the remainder-column
# classes (AMX_CONV_REM): FETCH_SIZE per launch, x2-corrected as in MI355X_MICROARCH.md (one PMC pass, kernel trace only).
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
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
for m in 0 1; do
  rm -rf /root/repo/gpurun_out/pmcrem_$m
  AMX_CONV_REM=$m timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/pmcrem_$m -o pmc --output-format csv -- python /tmp/dil16.py > /root/repo/gpurun_out/pmcrem_$m.log 2>&1
done
cd /root/repo
python - <<'PY' | tee gpurun_out/r04_rem_fetch.txt
import csv, glob, re, collections
print("# FETCH_SIZE (x2-corrected, MB per launch) of the last dilnet forward's 3x3 launches in order: input of the 512^2 layers = 16 x 512^2 x 52 ch x 4 B = 872 MB (28 ch: 470 MB)")
for m in (0, 1):
    path = glob.glob(f"gpurun_out/pmcrem_{m}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "FETCH_SIZE"]
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(int(r["Dispatch_Id"]), [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), 0.0])[1] += float(r["Counter_Value"])
    disp = list(per.values())
    last = [i for i, d in enumerate(disp) if "conv1_fwd_kernel" in d[0]][-1]
    print(f"AMX_CONV_REM={m}")
    for name, v in disp[last:]:
        if "conv_fwd_kernel<9" in name: print(f"   {name[:62]:62s} fetched {2 * v * 1024 / 1e6 / 1:8.0f} MB")
PY
rm -rf gpurun_out/pmcrem_0 gpurun_out/pmcrem_1
