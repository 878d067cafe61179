set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r02.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_r02.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
cat gpurun_out/bench_cur.json
LMH_DIAG=1 timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r02_phase_stamps.txt 2>&1
LMH_DIAG=1 LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r02_phase_stamps_nw2.txt 2>&1
