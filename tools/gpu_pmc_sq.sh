#!/bin/bash
# SQ issue / MFMA / LDS counters of the training step's kernels: two PMC passes (kernel trace only, serial schedule so that
# a kernel's counters are its own), summarised by tools/summarize_pmc_sq.py into profiles/<round>_pmc_sq.md.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 1 --serial --no-cpu-baseline --no-kernel-timing --no-extra --sustain-seconds 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace -d /root/repo/gpurun_out/pmc_sq -o pmc --output-format csv -- python /root/repo/bench.py $ARGS > /root/repo/gpurun_out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d /root/repo/gpurun_out/pmc_sq2 -o pmc --output-format csv -- python /root/repo/bench.py $ARGS > /root/repo/gpurun_out/pmc_sq2.log 2>&1
cd /root/repo; tail -2 gpurun_out/pmc_sq.log; tail -2 gpurun_out/pmc_sq2.log; python tools/summarize_pmc_sq.py ${2:-r03} ${1:-unknown} | head -40
