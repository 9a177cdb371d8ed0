#!/bin/bash
# GPU box: `agg` as q16 rows (option agg_q16 / env AG_AGG_Q16) against the default, same library, interleaved: bench lines per workload, then the
# deviation sweep.   tools/ab_aggq16.sh [reps=3] [fuzz cases=5000]
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
REPS=${1:-3}; CASES=${2:-5000}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %8.0f graph-steps/s  ' % ('$1', d['value']) + '  '.join('%s %.4f' % (n[:6], v['ms_per_launch']) for n,v in d['kernels'].items()))"; }
for m in "rope 256 10" "granular 128 10" "cloth 64 20"; do set -- $m
for i in $(seq $REPS); do for q in 0 1; do
  AG_AGG_Q16=$q python bench.py --material $1 --batch $2 --rollout-steps $3 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra 2>/dev/null | line "$1 agg_q16=$q"
done; done; done
echo "# deviation from the oracle, precision 2 only, same cases (tools/fuzz_parity.py $CASES 606 2):"
for q in 0 1; do echo -n "agg_q16=$q: "; AG_AGG_Q16=$q python tools/fuzz_parity.py $CASES 606 2 2>/dev/null | tail -2; done
echo "# forward goldens (tools/fwd_err.py), fast column:"
for q in 0 1; do echo "agg_q16=$q:"; AG_AGG_Q16=$q python tools/fwd_err.py 2>/dev/null | grep "^prec 2"; done
