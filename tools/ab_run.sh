#!/bin/bash
# GPU box: per-kernel forward timing (tools/time_forward.py, rope-1k batch 256) for every ab/lib*.so variant.
# Usage: tools/ab_run.sh [precision=2] [reps=20]; env AB_PMC="tag ..." adds an SQ counter pass for those variants.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
for so in ab/lib*.so; do
  tag=$(basename $so .so); tag=${tag#lib}
  printf "%-12s " $tag; AG_LIB_PATH=$REPO/$so python tools/time_forward.py ${1:-2} ${2:-20} 2>&1 | tail -1 | sed 's/^[^{]*//'
done
cd /tmp && export TMPDIR=/tmp
for tag in $AB_PMC; do
  OUT=$REPO/gpurun_out/pmc_ab_$tag; rm -rf $OUT; mkdir -p $OUT
  AG_LIB_PATH=$REPO/ab/lib$tag.so timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
     -d $OUT -o pmc -- python $REPO/tools/time_forward.py ${1:-2} 3 > $OUT/log.txt 2>&1
  echo "== PMC $tag"; python $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*.db") 2>/dev/null | grep -A9 "edge_encode"
done
