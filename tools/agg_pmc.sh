#!/bin/bash
# GPU box: PMC passes over tools/agg_check.py (both segment-reduce kernels run in it).  Usage: tools/agg_pmc.sh [lib.so] [material n_obj batch]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
LIBP=${1:-}; shift
ARGS=${@:-rope 1000 256 3}
mkdir -p $REPO/gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/aggpmc_$i; rm -rf $OUT; mkdir -p $OUT
  AG_LIB_PATH=$LIBP timeout 300 rocprofv3 --pmc $CTRS -d $OUT -o pmc -- python $REPO/tools/agg_check.py $ARGS > $OUT/log.txt 2>&1 || tail -3 $OUT/log.txt
done
python $REPO/tools/rocpd_summary.py pmc $(find $REPO/gpurun_out/aggpmc_* -name "*.db") 2>/dev/null | grep -A48 -E "^aggregate" | tee $REPO/gpurun_out/agg_pmc.txt
