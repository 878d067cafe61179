set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile_rollout.sh r02_c3 > gpurun_out/prof_r02_c3.log 2>&1
bash scripts/profile_rollout.sh r02_c2 --config 2 > gpurun_out/prof_r02_c2.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/bench_r02_c3.json 2> gpurun_out/bench_r02_c3.err
timeout -k 10 600 python bench.py --config 2 > gpurun_out/bench_r02_c2.json 2> gpurun_out/bench_r02_c2.err
LMH_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 2 --backend gloo --no-cpu-baseline > gpurun_out/bench_2rank.log 2> gpurun_out/bench_2rank.err
grep '^{"metric"' gpurun_out/bench_2rank.log > gpurun_out/bench_r02_2rank_gloo.json
LMH_DIAG=1 timeout -k 10 300 python scripts/gpu_barrier_share.py 3 40 200 > gpurun_out/r02_barrier_share.txt 2>&1
LMH_DIAG=1 timeout -k 10 300 python scripts/gpu_barrier_share.py 2 10 100 >> gpurun_out/r02_barrier_share.txt 2>&1
LMH_DIAG=1 LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/gpu_wave_timeline.py 3 130 > gpurun_out/tl_ds.txt 2>&1
LMH_DIAG=1 LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/gpu_wave_timeline.py 3 220 > gpurun_out/tl_ss.txt 2>&1
LMH_DIAG=1 timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r02_phase_stamps.txt 2>&1
LMH_DIAG=1 LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r02_phase_stamps_nw2.txt 2>&1
timeout -k 10 600 python scripts/precision_sweep.py 1024 600 gpurun_out/r02_precision_sweep.json > gpurun_out/sweep.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
tail -1 gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r02.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_r02.log
