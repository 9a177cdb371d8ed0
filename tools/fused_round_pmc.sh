#!/bin/bash
# GPU box: one propagation round as two launches (aggregate_half_kernel + node_update_kernel, AG_FUSE_AGG=0, shipped) vs ONE launch
# (node_update_kernel<.., FUSE>, AG_FUSE_AGG=2): launch times, end-to-end, HBM bytes and SQ counters per dispatch (separate --pmc passes).
#   tools/fused_round_pmc.sh > gpurun_out/r04_fused_round.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
echo "# per-launch ms of one C2 forward (tools/time_forward.py; with fuse 2 the reduce is inside node_update)"
for f in 0 2; do echo -n "AG_FUSE_AGG=$f  "; AG_FUSE_AGG=$f python tools/time_forward.py 2 20 2>&1 | tail -1; done
echo "# bench.py (two rollout streams), graph-steps/s"
for i in 1 2; do for f in 0 2; do
  AG_FUSE_AGG=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('AG_FUSE_AGG=$f %8.0f  ' % d['value'] + '  '.join('%s %.4f' % (n[:9], v['ms_per_launch']) for n,v in d['kernels'].items()))"
done; done
cd /tmp && export TMPDIR=/tmp
for f in 0 2; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
    OUT=$REPO/gpurun_out/fr_$f; rm -rf $OUT; mkdir -p $OUT
    AG_FUSE_AGG=$f timeout 300 rocprofv3 --pmc $grp -d $OUT -o pmc -- python $REPO/tools/time_forward.py 2 3 > /dev/null 2>&1
    echo "== AG_FUSE_AGG=$f counters: $grp"
    python $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*.db") 2>/dev/null | grep -A9 -E "^(aggregate_half_kernel|node_update_kernel)" | grep -v "^--"
    rm -rf $OUT
  done
done
