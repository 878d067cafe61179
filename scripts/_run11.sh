mkdir -p gpurun_out
timeout -k 5 60 scripts/microbench/issue_cost
LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/gpu_wave_timeline.py 3 130 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/wave_timeline_c.log; cat gpurun_out/wave_timeline_c.log
bash scripts/profile_rollout.sh r02_c3 > gpurun_out/prof_r02_c3.log 2>&1; tail -3 gpurun_out/prof_r02_c3.log
bash scripts/profile_rollout.sh r02_c2 --config 2 > gpurun_out/prof_r02_c2.log 2>&1; tail -3 gpurun_out/prof_r02_c2.log
