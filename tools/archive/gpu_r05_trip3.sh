cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python tools/gpu_predict_knobs.py ) > gpurun_out/r05_predict_knobs.log 2>&1
( AMX_RVAE_NO_AB=1 timeout 300 python tools/gpu_env_rvae.py ) > gpurun_out/r05_rvae_mt_ab.log 2>&1
tail -8 gpurun_out/r05_predict_knobs.log; tail -6 gpurun_out/r05_rvae_mt_ab.log
