#!/bin/bash
# LDS bank-conflict counters of the conv kernels' LDS address patterns in isolation (tools/micro/lds_b128_probe.hip),
# one PMC pass (kernel trace only), summarised into profiles/<round>_lds_conflicts.md.   tools/gpu_pmc_lds.sh r06 <head>
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
[ -x tools/micro/lds_b128_probe.bin ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/micro/lds_b128_probe.bin tools/micro/lds_b128_probe.hip
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d /root/repo/gpurun_out/pmc_lds -o pmc --output-format csv -- /root/repo/tools/micro/lds_b128_probe.bin > /root/repo/gpurun_out/pmc_lds.log 2>&1
cd /root/repo; tail -3 gpurun_out/pmc_lds.log
python tools/summarize_pmc_lds.py "$1" "$2"
