#!/bin/bash
# Register / occupancy table of every kernel instantiation in one source: tools/kernel_resources.sh conv_fwd [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I atomai_amd/csrc -I include "$@" \
  -c atomai_amd/csrc/$f.hip -o /tmp/kres_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c '
import re,sys,subprocess
rows=[];cur=None
for ln in sys.stdin:
    m=re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]):\s+(\S+)",ln)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        name=subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip()
        cur={"name":re.sub(r"\(.*","",name).replace("void ","")}; rows.append(cur)
    else: cur[k.split()[0]]=v
for r in rows:
    print("%-62s sgpr %4s vgpr %4s agpr %4s scratch %4s occ %s"%(r["name"][:62],r.get("TotalSGPRs"),r.get("VGPRs"),r.get("AGPRs"),r.get("ScratchSize"),r.get("Occupancy")))
'
