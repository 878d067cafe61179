# Runs ON THE GPU BOX (gpurun -- bash scripts/gpu_evidence_bundle.sh [part]): the round's measurements, written under gpurun_out/.
# part = prof | bench | diag | tests | all (default); the parts are independent so that one gpurun call stays inside its time limit.
set -e
PART=${1:-all}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ $PART = prof ] || [ $PART = all ]; then
  bash scripts/profile_rollout.sh r04_c3 > gpurun_out/prof_r04_c3.log 2>&1
  bash scripts/profile_rollout.sh r04_c2 --config 2 > gpurun_out/prof_r04_c2.log 2>&1
  cd $GRAFT_REPO_ROOT
fi
if [ $PART = bench ] || [ $PART = all ]; then
  timeout -k 10 600 python bench.py > gpurun_out/bench_r04_c3.json 2> gpurun_out/bench_r04_c3.err
  timeout -k 10 600 python bench.py --config 2 > gpurun_out/bench_r04_c2.json 2> gpurun_out/bench_r04_c2.err
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r04_c3_steps20.json 2> gpurun_out/bench_r04_c3_steps20.err   # the driver's flags
  timeout -k 10 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_r04_c5.json 2> gpurun_out/bench_r04_c5.err
  timeout -k 10 600 python bench.py --coupled --no-cpu-baseline > gpurun_out/bench_r04_c3_coupled.json 2> gpurun_out/bench_r04_c3_coupled.err
  timeout -k 10 600 python bench.py --host-io --no-cpu-baseline > gpurun_out/bench_r04_c3_hostio.json 2> gpurun_out/bench_r04_c3_hostio.err
  timeout -k 10 600 python bench.py --config 2 --push 1.0 --no-cpu-baseline > gpurun_out/bench_r04_c2_push1.json 2> gpurun_out/bench_r04_c2_push1.err || true   # BASELINE's amplitude: robots fall, rc 3
  timeout -k 10 600 python bench.py --mode eval > gpurun_out/bench_r04_eval.json 2> gpurun_out/bench_r04_eval.err
  LMH_BENCH_DEVICE=0 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --no-cpu-baseline > gpurun_out/bench_2rank.log 2> gpurun_out/bench_2rank.err
  grep '^{"metric"' gpurun_out/bench_2rank.log > gpurun_out/bench_r04_2rank_gloo.json
fi
if [ $PART = diag ] || [ $PART = all ]; then
  export LMH_DIAG=1
  timeout -k 10 300 python scripts/diag.py barrier 3 1000 1000 > gpurun_out/r04_barrier_share.txt 2>&1
  timeout -k 10 300 python scripts/diag.py barrier 3 200 1300 >> gpurun_out/r04_barrier_share.txt 2>&1
  timeout -k 10 300 python scripts/diag.py barrier 3 200 1100 >> gpurun_out/r04_barrier_share.txt 2>&1
  timeout -k 10 300 python scripts/diag.py barrier 3 200 1500 >> gpurun_out/r04_barrier_share.txt 2>&1
  timeout -k 10 300 python scripts/diag.py barrier 2 200 600 >> gpurun_out/r04_barrier_share.txt 2>&1
  timeout -k 10 300 python scripts/diag.py barrier 5 200 600 >> gpurun_out/r04_barrier_share.txt 2>&1
  LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/diag.py timeline 3 1120 > gpurun_out/tl_ds.txt 2>&1
  LMH_DIAG_NW2=1 timeout -k 10 200 python scripts/diag.py timeline 3 1300 > gpurun_out/tl_ss.txt 2>&1
  timeout -k 10 200 python scripts/gpu_phase_stamps.py 1024 1 40 > gpurun_out/r04_phase_stamps.txt 2>&1
  unset LMH_DIAG
  # production-kernel timeline (variant tl1 = -DLMH_SUBSTAMPS -DLMH_DIAG_TL=1, built beside the diagnostic library: see profiles/README.md) and the free sets along the gait
  : > gpurun_out/r04_prod_timeline.txt
  for t in 1120 1350 1550; do LMH_VARIANT=tl1 timeout -k 10 200 python scripts/diag.py ptimeline 3 $t >> gpurun_out/r04_prod_timeline.txt 2>&1; echo >> gpurun_out/r04_prod_timeline.txt; done
  LMH_VARIANT=tl1 timeout -k 10 200 python scripts/diag.py ptimeline 2 600 1024 >> gpurun_out/r04_prod_timeline.txt 2>&1
  timeout -k 10 300 python scripts/route_probe.py 3 1024 > gpurun_out/r04_route_probe.txt 2>&1
  timeout -k 10 300 python scripts/diag.py rounds 3 4000 400 > gpurun_out/r04_qp_rounds.txt 2>&1
  timeout -k 10 300 python scripts/diag.py rounds 2 2000 200 >> gpurun_out/r04_qp_rounds.txt 2>&1
  timeout -k 10 600 python scripts/precision_sweep.py 1024 600 gpurun_out/r04_precision_sweep.json > gpurun_out/sweep.log 2>&1
fi
if [ $PART = tests ] || [ $PART = all ]; then
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  tail -1 gpurun_out/smoke.log
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r04.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r04.log; exit 1; }
  tail -3 gpurun_out/pytest_gpu_r04.log
fi
