set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r02.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_r02.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
python -c "
import json,sys; r=json.load(open('gpurun_out/bench_cur.json')); print('c3', r['value'], r['roofline']['kernel_ms'], r['instances_flagged'])"
done
LMH_DIAG=1 timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r02_phase_stamps.txt 2>&1
grep -E "refs|total" gpurun_out/r02_phase_stamps.txt
