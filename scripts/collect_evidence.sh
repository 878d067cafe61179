#!/bin/bash
# After `gpurun -- bash scripts/gpu_evidence_bundle.sh`: summarise the rocprofv3 passes and copy the round's evidence from gpurun_out/ into profiles/.
# Every piece is optional: what a partial bundle did not produce is skipped.
cd "$(dirname "$0")/.."
for t in r04_c3 r04_c2; do [ -d gpurun_out/prof_$t ] && python scripts/summarise_profile.py $t > /dev/null; done
for f in bench_r04_c3 bench_r04_c2 bench_r04_c5 bench_r04_c3_coupled bench_r04_c3_hostio bench_r04_2rank_gloo bench_r04_c2_push1 bench_r04_eval bench_r04_c3_steps20; do [ -s gpurun_out/$f.json ] && cp gpurun_out/$f.json profiles/$f.json; done
[ -s gpurun_out/r04_barrier_share.txt ] && grep -v amdgpu gpurun_out/r04_barrier_share.txt > profiles/r04_barrier_share.txt
[ -s gpurun_out/r04_prod_timeline.txt ] && grep -v amdgpu gpurun_out/r04_prod_timeline.txt > profiles/r04_prod_timeline.txt
[ -s gpurun_out/r04_route_probe.txt ] && grep -v amdgpu gpurun_out/r04_route_probe.txt > profiles/r04_route_probe.txt
[ -s gpurun_out/r04_qp_rounds.txt ] && grep -v amdgpu gpurun_out/r04_qp_rounds.txt > profiles/r04_qp_rounds.txt
if [ -s gpurun_out/tl_ds.txt ]; then
  { echo "== evaluation after 1120 ticks (LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/diag.py timeline 3 1120; the first line names the support phase) =="; grep -v amdgpu gpurun_out/tl_ds.txt; echo
    echo "== evaluation after 1300 ticks (python scripts/diag.py timeline 3 1300) =="; grep -v amdgpu gpurun_out/tl_ss.txt; } > profiles/r04_wave_timeline.txt
fi
[ -s gpurun_out/r04_phase_stamps.txt ] && grep -v amdgpu.ids gpurun_out/r04_phase_stamps.txt > profiles/r04_phase_stamps.txt
[ -s gpurun_out/r04_precision_sweep.json ] && cp gpurun_out/r04_precision_sweep.json profiles/
[ -s gpurun_out/pytest_gpu_r04.log ] && cp gpurun_out/pytest_gpu_r04.log profiles/r04_pytest_gpu.log
ls profiles | grep r04
