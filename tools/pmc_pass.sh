#!/bin/bash
# One rocprofv3 PMC pass over a short bench run.  Usage: tools/pmc_pass.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CTRS -d $OUT -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra $* > $OUT/log.txt 2>&1
python $REPO/tools/rocpd_summary.py pmc $OUT/pmc_results.db 2>/dev/null | head -${PMC_LINES:-60}
