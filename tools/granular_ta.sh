#!/bin/bash
# GPU box: why is the segment reduce's HBM fraction lower on granular-2k than on rope / cloth?  Texture-path counters of the reduce, one
# counter group per pass (no trace domains), rope C2 and granular-2k x 128, one rollout stream and (granular) the engine's default two.
#   tools/granular_ta.sh  ->  gpurun_out/granular_ta.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUTF=$REPO/gpurun_out/granular_ta.txt; : > $OUTF
cd /tmp && export TMPDIR=/tmp
for cfg in "rope 256 1" "granular 128 1" "granular 128 2"; do set -- $cfg
  dbs=""
  i=0
  for CTRS in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TD_TC_STALL_sum" "TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); OUT=$REPO/gpurun_out/gta_$1_$3_$i; rm -rf $OUT; mkdir -p $OUT
    timeout 200 rocprofv3 --pmc $CTRS -d $OUT -o pmc -- python $REPO/bench.py --material $1 --batch $2 --rollout-steps 10 --steps 1 --warmup 1 --streams $3 \
        --no-cpu-baseline --no-profile --no-extra > $OUT/log.txt 2>&1 || echo "pass failed: $1 $3 $CTRS" >> $OUTF
    dbs="$dbs $(find $OUT -name '*.db' | head -1)"
  done
  echo "## $1 x $2, rollout_streams $3: aggregate_half_kernel, mean per dispatch" >> $OUTF
  python $REPO/tools/rocpd_summary.py pmc $dbs 2>/dev/null | grep -A14 -E "^aggregate_half" >> $OUTF
  rm -rf $REPO/gpurun_out/gta_$1_$3_*
done
cat $OUTF
