#!/bin/bash
# rocprofv3 kernel trace of the MPPI iteration (bench_mpc.py), shared-state rollout on / off.  Usage: tools/profile_mpc.sh <tag> [bench_mpc args]
set -u
TAG=${1:-r06}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_mpc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ss in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_ss$ss -o trace -- python $REPO/bench_mpc.py --steps 3 --warmup 1 --shared-state $ss "$@" > $OUT/trace_ss$ss.log 2>&1
  db=$(find $OUT/trace_ss$ss -name "*.db" | head -1)
  python $REPO/tools/rocpd_summary.py trace $db > $OUT/mpc_trace_ss$ss.txt 2>&1
  python $REPO/tools/rocpd_summary.py streams $db 400 > $OUT/mpc_streams_ss$ss.txt 2>&1; rm -rf $OUT/trace_ss$ss
done
head -45 $OUT/mpc_trace_ss1.txt
