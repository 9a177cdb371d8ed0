#!/bin/bash
# GPU box: A/B the segment reduce as its own launch (AG_FUSE_AGG=0) vs fused into node_update through LDS (AG_FUSE_AGG=2).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for f in 0 2; do AG_FUSE_AGG=$f python tools/time_forward.py 2 20 2>&1 | tail -1; done
for i in 1 2; do for f in 0 2; do
  AG_FUSE_AGG=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fuse=$f %8.0f graph-steps/s  ' % d['value'] + '  '.join('%s %.4f' % (n[:6], v['ms_per_launch']) for n,v in d['kernels'].items()))"
done; done
