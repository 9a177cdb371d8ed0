#!/bin/bash
# GPU box: end-to-end A/B of the rollout configuration around the current kernels: segment reduce separate (AG_FUSE_AGG=0) or inside
# node_update (2), and 1..4 batch parts on separate streams (bench.py --streams).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2; do for f in 0 2; do for st in 1 2 3 4; do
  AG_FUSE_AGG=$f python bench.py --steps 5 --warmup 2 --streams $st --no-cpu-baseline --no-extra --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fuse=$f streams=$st %8.0f graph-steps/s' % d['value'])"
done; done; done
