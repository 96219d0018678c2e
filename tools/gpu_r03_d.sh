#!/bin/bash
# Round-3 trip D: predictor pipeline with the collect worker thread — tests, 256-frame and 4096-frame runs, host trace
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_seg_gpu.py tests/test_locator_gpu.py -m gpu -q -x -k "predict or locat or config3" 2>&1 | tail -8 ) > gpurun_out/r03d_pytest.log 2>&1
( timeout 900 python tools/bench_extra.py predict predict4096 ) > gpurun_out/r03d_predict.log 2>&1
( AMX_PREDICT_TRACE=1 timeout 600 python - <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_extra as bx
bx.bench_predict_full(frames=640)
PY
) > gpurun_out/r03d_trace.log 2>&1
echo "== pytest"; tail -4 gpurun_out/r03d_pytest.log; echo "== predict"; grep "^{" gpurun_out/r03d_predict.log | cut -c1-700; echo "== trace"; grep chunk gpurun_out/r03d_trace.log | sed -n 20,36p
