cd /root/repo; export TMPDIR=/tmp
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
for m in 0 3; do
  AMX_CONV_XCD=$m timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/pmcxcd_$m -o pmc --output-format csv -- python /tmp/dil16.py > /root/repo/gpurun_out/pmcxcd_$m.log 2>&1
done
cd /root/repo
python - <<'PY' | tee gpurun_out/r03_xcd_fetch.txt
import csv, glob, re, collections
for m in (0, 3):
    path = glob.glob(f"gpurun_out/pmcxcd_{m}/**/*counter_collection.csv", recursive=True)[0]
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        d.setdefault(re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), []).append(float(r["Counter_Value"]))
    print(f"AMX_CONV_XCD={m}")
    for k, v in d.items():
        if "conv_fwd_kernel<9" in k: print(f"   {k[:58]:58s} fetched {2 * sum(v[-2:]) / 2 * 1024 / 1e6:7.0f} MB / launch (x2-corrected)")
PY
