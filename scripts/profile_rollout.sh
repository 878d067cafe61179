#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of a bench workload.
# Counters are collected in their own passes (never combined with a trace domain).  Output: gpurun_out/prof_$1/
# then summarised by scripts/summarise_profile.py into profiles/.   Usage: profile_rollout.sh TAG [bench.py args, e.g. --config 2]
set -e
TAG=${1:-r04_c3}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf $O
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/kt.log 2>&1
# PMC passes: the same command with fewer steps (one launch = one whole rollout segment, ~1.5 s); each pass on its own, a counter the
# box does not list fails its pass only (the summariser reports what exists)
P="--steps ${PMC_STEPS:-3} --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $B $P > $O/fetch.log 2>&1 || echo "pass fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $B $P > $O/write.log 2>&1 || echo "pass write failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d $O/sq -o sq -- $B $P > $O/sq.log 2>&1 || echo "pass sq failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $O/sq2 -o sq2 -- $B $P > $O/sq2.log 2>&1 || echo "pass sq2 failed"
# fp64 instruction classes (a counted flop figure beside SURVEY's nominal one)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS \
    --output-format csv -d $O/f64 -o f64 -- $B $P > $O/f64.log 2>&1 || echo "pass f64 failed"
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH \
    --output-format csv -d $O/ins -o ins -- $B $P > $O/ins.log 2>&1 || echo "pass ins failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o grbm -- $B $P > $O/grbm.log 2>&1 || echo "pass grbm failed"
grep -h '^{"metric"' $O/kt.log > $O/bench_line.json || true
find $O -name "*_kernel_trace.csv" -delete      # per-dispatch trace is large; the stats table is what is kept
ls -R $O | head -40
