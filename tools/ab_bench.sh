#!/bin/bash
# GPU box: bench.py (2-stream rollout, C2) for every ab/lib*.so variant, REPS times each, interleaved.  Usage: tools/ab_bench.sh [reps=2] [bench args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
REPS=${1:-2}; shift
for i in $(seq $REPS); do
for so in ab/lib*.so; do
  tag=$(basename $so .so); tag=${tag#lib}
  AG_LIB_PATH=$REPO/$so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %8.0f graph-steps/s  ' % ('$tag', d['value']) + '  '.join('%s %.4f' % (n[:6], v['ms_per_launch']) for n,v in d['kernels'].items()))"
done; done
