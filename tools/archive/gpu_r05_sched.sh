cd /root/repo; mkdir -p gpurun_out
( timeout 600 python tools/gpu_step_ab.py "" "lib=ilp" "lib=mmc" ) > gpurun_out/r05_sched_strategy_step_ab.log 2>&1
( timeout 300 python tools/gpu_rvae_ab.py ilp mmc ) > gpurun_out/r05_sched_strategy_rvae_ab.log 2>&1
tail -4 gpurun_out/r05_sched_strategy_step_ab.log | cut -c1-250; tail -5 gpurun_out/r05_sched_strategy_rvae_ab.log
