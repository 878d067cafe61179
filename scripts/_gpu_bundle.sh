set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "torchrun" > gpurun_out/pytest_one.log 2>&1 || { tail -40 gpurun_out/pytest_one.log; exit 1; }
tail -3 gpurun_out/pytest_one.log
