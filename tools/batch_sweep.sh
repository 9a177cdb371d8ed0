#!/bin/bash
# GPU box: per-graph cost vs batch for one workload (bench.py kernel breakdown).  Usage: tools/batch_sweep.sh cloth 20 "64 128 256 512" [extra bench args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
MAT=$1; T=$2; BATCHES=$3; shift 3
for b in $BATCHES; do
  python bench.py --material $MAT --batch $b --rollout-steps $T --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=$b
k=d['kernels']
print('batch %4d  %8.0f graph-steps/s  %7.2f ms/pass | per-launch us per graph: ' % (b, d['value'], d['ms_per_step']) + '  '.join('%s %.2f' % (n[:6], v['ms_per_launch']*1e3/b) for n,v in k.items()))"
done
