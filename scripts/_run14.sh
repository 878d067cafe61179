mkdir -p gpurun_out
python -m pytest tests/test_gpu_round2.py -m gpu -q -k plant 2>&1 | tail -3
bash scripts/profile_rollout.sh r02_c3 > gpurun_out/prof_r02_c3.log 2>&1; tail -2 gpurun_out/prof_r02_c3.log
bash scripts/profile_rollout.sh r02_c2 --config 2 > gpurun_out/prof_r02_c2.log 2>&1; tail -2 gpurun_out/prof_r02_c2.log
python bench.py > gpurun_out/bench_r02_c3.json 2> gpurun_out/bench_r02_c3.err; echo "rc $?"; cat gpurun_out/bench_r02_c3.json | cut -c1-400
python bench.py --config 2 > gpurun_out/bench_r02_c2.json 2> gpurun_out/bench_r02_c2.err; echo "rc $?"
LMH_BENCH_DEVICE=0 python bench.py --gpus 2 --backend gloo > gpurun_out/bench_r02_2rank_gloo.json 2> gpurun_out/bench_r02_2rank_gloo.err; echo "rc $?"; cat gpurun_out/bench_r02_2rank_gloo.json | cut -c1-300
LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/gpu_wave_timeline.py 3 130 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_wave_timeline.txt
LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/gpu_wave_timeline.py 3 250 1024 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02_wave_timeline.txt
LMH_DIAG=1 python scripts/gpu_barrier_share.py 3 40 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_barrier_share.txt
LMH_DIAG=1 python scripts/gpu_barrier_share.py 2 10 100 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02_barrier_share.txt
cat gpurun_out/r02_barrier_share.txt
