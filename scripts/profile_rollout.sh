#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of a bench workload.
# Counters are collected in their own passes (never combined with a trace domain).  Output: gpurun_out/prof_$1/
# then summarised by scripts/summarise_profile.py into profiles/.   Usage: profile_rollout.sh TAG [bench.py args, e.g. --config 2]
set -e
TAG=${1:-r02_c3}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf $O
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $B --steps 5 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $B --steps 5 --warmup 1 > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d $O/sq -o sq -- $B --steps 5 --warmup 1 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $O/sq2 -o sq2 -- $B --steps 5 --warmup 1 > $O/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o grbm -- $B --steps 5 --warmup 1 > $O/grbm.log 2>&1
grep -h '^{"metric"' $O/kt.log > $O/bench_line.json || true
find $O -name "*_kernel_trace.csv" -delete      # per-dispatch trace is large; the stats table is what is kept
ls -R $O | head -40
