#!/bin/bash
# GPU box: texture-addresser / data-return counters of the segment reduce (one counter group per pass).  tools/agg_pmc_ta.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TA_TOTAL_WAVEFRONTS_sum" "TA_FLAT_READ_WAVEFRONTS_sum" "TD_TD_BUSY_sum" "TD_TC_STALL_sum" "TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TD_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum" "TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/aggta_$i; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --pmc $CTRS -d $OUT -o pmc -- python $REPO/tools/time_forward.py 2 3 > $OUT/log.txt 2>&1 || echo "pass failed: $CTRS"
done
python $REPO/tools/rocpd_summary.py pmc $(find $REPO/gpurun_out/aggta_* -name "*.db") 2>/dev/null | grep -A20 -E "^(aggregate_half|node_update_kernel<PrecB3, false)" | tee $REPO/gpurun_out/agg_pmc_ta.txt
