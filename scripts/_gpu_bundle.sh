set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spec in "1024 40" "2048 40" "4096 40" "8192 40" "4000 40"; do
set -- $spec
timeout -k 10 300 python bench.py --no-cpu-baseline --instances $1 --ticks $2 --steps 12 --warmup 3 > gpurun_out/bench_n.json 2> gpurun_out/bench_n.err
python -c "
import json,sys; r=json.load(open('gpurun_out/bench_n.json')); print('n=$1 ticks=$2', round(r['value']), round(r['roofline']['kernel_ms'],3), r['instances_flagged'])"
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r02.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_r02.log
