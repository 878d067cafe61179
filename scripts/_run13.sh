mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
python bench.py --config 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
