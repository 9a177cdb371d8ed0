#!/bin/bash
# GPU box: one rollout stream against two (bench.py --streams) per workload, REPS times each, interleaved.  Usage: tools/ab_streams2.sh [reps=2]
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { tag=$1; shift
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s %8.0f graph-steps/s  %8.3f ms/pass' % ('$tag', d['value'], d['ms_per_step']))"
}
for rep in $(seq ${1:-2}); do
for s in 1 2; do
  one "rope-1k x256 s$s" --streams $s
  one "rope-1k x64 s$s" --streams $s --batch 64
  one "rope-1k x1024 s$s" --streams $s --batch 1024
  one "granular-2k x128 s$s" --streams $s --material granular --batch 128
  one "cloth-4k x64 x20 s$s" --streams $s --material cloth --batch 64 --rollout-steps 20
done; done
