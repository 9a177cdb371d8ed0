#!/bin/bash
# GPU box: the shipped non-temporal hints (ab/libN.so) against plain loads / stores (ab/libbase.so) — first: tools/ab_build.sh N="" base="-DAG_NO_NT" —
# at C2 with one and two rollout streams, on granular-2k and cloth-4k, and in the f32 / split-bf16 modes (profiles/r05_nt_hints.txt, round 4; the rev0..3 rows there came from a temporary switch for the direction of the reduce's walk).
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { # tag lib extra-args...
  tag=$1; so=$2; shift 2
  AG_LIB_PATH=$PWD/$so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s %8.0f graph-steps/s  ' % ('$tag', d['value']) + '  '.join('%s %.4f' % (n[:6], v['ms_per_launch']) for n,v in d['kernels'].items()))"
}
for rep in 1 2; do
one "N s1" ab/libN.so --streams 1
one "base s1" ab/libbase.so --streams 1
done
one "N s2" ab/libN.so --streams 2
one "base s2" ab/libbase.so --streams 2
for mat in granular cloth; do
  if [ $mat = granular ]; then A="--material granular --batch 128"; else A="--material cloth --batch 64 --rollout-steps 20"; fi
  for rep in 1 2; do one "base $mat" ab/libbase.so $A; one "N $mat" ab/libN.so $A; done
done
for prec in f32 bf16x3; do one "base $prec" ab/libbase.so --precision $prec; one "N $prec" ab/libN.so --precision $prec; done
