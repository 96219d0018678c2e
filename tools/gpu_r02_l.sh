#!/bin/bash
# Round-2 GPU trip L: lattice-mode weight gradients of the dilated layers: gpu tests, dilnet training A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r02l_pytest_gpu.log 2>&1
for v in 0 1; do
  ( AMX_CONV_LATTICE=$v timeout 600 python -c "
import sys; sys.path.insert(0, 'tools')
import bench_extra as B
B.bench_segfamily(models=('dilnet',))
" ) > gpurun_out/r02l_dilnet_train_lat$v.log 2>&1
done
echo "== pytest"; tail -3 gpurun_out/r02l_pytest_gpu.log; for v in 0 1; do echo "== lattice $v"; grep '^{' gpurun_out/r02l_dilnet_train_lat$v.log; done
