mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for v in "" libm gjsi licm; do
LMH_VARIANT=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] c3', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
LMH_VARIANT=$v python bench.py --config 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] c2', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
done
LMH_VARIANT=gjsi python -m pytest tests -m gpu -q -x 2>&1 | tail -3
