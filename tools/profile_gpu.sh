#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra $*"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
BENCH1="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra $*"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $BENCH1 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH1 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH1 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
ls -la $OUT $OUT/*/* | head -60
