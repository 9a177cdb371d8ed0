#!/bin/bash
# GPU box: SQ/GRBM counters of one forward (tools/time_forward.py) for a library variant.
# Usage: tools/pmc_fwd.sh <tag> <lib.so|default> [env assignments...]   (prints the edge-encoder block)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; LIB=$2; shift 2
OUT=$REPO/gpurun_out/pmc_fwd_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ "$LIB" != default ] && export AG_LIB_PATH=$REPO/$LIB
env "$@" timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
   -d $OUT -o pmc -- python $REPO/tools/time_forward.py ${PREC:-2} 3 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | sed 's/^[^{]*//'
echo "== PMC $TAG"; python $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*.db") 2>/dev/null | grep -A9 "^edge_encode"
