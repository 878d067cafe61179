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
timeout -k 10 300 python bench.py --no-cpu-baseline --config 2 > gpurun_out/bench_cur2.json 2> gpurun_out/bench_cur2.err
python -c "
import json,sys; r=json.load(open('gpurun_out/bench_cur2.json')); print('c2', r['value'], r['roofline']['kernel_ms'], r['instances_flagged'])"
LMH_DIAG=1 LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/gpu_wave_timeline.py 3 220 > gpurun_out/tl_ss.txt 2>&1
head -20 gpurun_out/tl_ss.txt
