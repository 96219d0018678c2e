#!/bin/bash
# Round-2 GPU trip F: coalesced epilogue + per-wave statistics: gpu tests, per-shape and step-level A/B, phase timing
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r02f_pytest_gpu.log 2>&1
( timeout 900 python tools/gpu_probe_r02.py unet dilnet ) > gpurun_out/r02f_probe.log 2>&1
( timeout 900 python tools/gpu_step_ab.py "" "lib=epi0" ) > gpurun_out/r02f_step_ab.log 2>&1
( timeout 600 python tools/gpu_conv_phases.py ) > gpurun_out/r02f_phases.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r02f_pytest_gpu.log; echo "== probe"; grep -v Warn gpurun_out/r02f_probe.log | tail -40; echo "== step"; grep -v Warn gpurun_out/r02f_step_ab.log | tail -3; echo "== phases"; grep -v Warn gpurun_out/r02f_phases.log | head -36
