mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
for v in "" licm; do
LMH_VARIANT=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] c3', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
LMH_VARIANT=$v python bench.py --config 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] c2', d['value'], d['roofline']['kernel_ms'], d['instances_flagged'])"
done
LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/gpu_wave_timeline.py 3 250 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/wave_timeline_gj.log; cat gpurun_out/wave_timeline_gj.log
