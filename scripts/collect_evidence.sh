#!/bin/bash
# After `gpurun -- bash scripts/gpu_evidence_bundle.sh`: summarise the rocprofv3 passes and copy the round's evidence from gpurun_out/ into profiles/.
set -e
cd "$(dirname "$0")/.."
python scripts/summarise_profile.py r02_c3 > /dev/null
python scripts/summarise_profile.py r02_c2 > /dev/null
for f in bench_r02_c3 bench_r02_c2 bench_r02_2rank_gloo; do cp gpurun_out/$f.json profiles/$f.json; done
grep -v amdgpu gpurun_out/r02_barrier_share.txt > profiles/r02_barrier_share.txt
{ echo "== double support (python scripts/gpu_wave_timeline.py 3 130, LMH_DIAG=1 LMH_DIAG_NW2=1) =="; grep -v amdgpu gpurun_out/tl_ds.txt; echo
  echo "== single support (python scripts/gpu_wave_timeline.py 3 220) =="; grep -v amdgpu gpurun_out/tl_ss.txt; } > profiles/r02_wave_timeline.txt
{ echo "== single-wave debug kernel, warm start, 40 rollout ticks first (LMH_DIAG=1 python scripts/gpu_phase_stamps.py 1024 1 40) =="; grep -v amdgpu.ids gpurun_out/r02_phase_stamps.txt; echo
  echo "== two-wave debug kernel (LMH_DIAG_NW2=1) =="; grep -v amdgpu.ids gpurun_out/r02_phase_stamps_nw2.txt | sed -n '/two-wave schedule/,$p'; } > profiles/r02_phase_stamps.txt
cp gpurun_out/r02_precision_sweep.json profiles/
cp gpurun_out/pytest_gpu_r02.log profiles/r02_pytest_gpu.log
