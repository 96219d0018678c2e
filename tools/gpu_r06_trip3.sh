#!/bin/bash
# round-6 third trip: x-packed lattice tiles A/B, rewritten covariance builder, full gpu test tier
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 700 python tools/gpu_step_ab.py "AMX_CONV_XPACK=1" "AMX_CONV_XPACK=0" > $O/r06_xpack_ab.log 2>&1; echo "xpack rc=$?"
timeout 600 python tools/bench_extra.py dkl > $O/r06_dkl.log 2>&1; echo "dkl rc=$?"
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r06_pytest_gpu_trip3.log 2>&1; echo "tests rc=$?"
tail -3 $O/r06_xpack_ab.log; tail -2 $O/r06_dkl.log | cut -c1-1200; tail -5 $O/r06_pytest_gpu_trip3.log
