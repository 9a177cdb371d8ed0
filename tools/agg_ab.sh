#!/bin/bash
# A/B of the segment reduce with / without self-edge elision on the three bench workloads (per-kernel HIP-event times of bench.py's roofline pass)
export TMPDIR=/tmp
run() { # label, env...
  label=$1; shift
  for mat in rope granular cloth; do
    b=256; t=10; [ $mat = granular ] && b=128; [ $mat = cloth ] && { b=64; t=20; }
    env "$@" python bench.py --material $mat --batch $b --rollout-steps $t --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '$mat', round(d['value'],1), {k:round(v['ms_per_launch'],4) for k,v in d['kernels'].items()})"
  done
}
run elide AG_X=0
run no_elision AG_SELF_EDGES=0
run elide AG_X=0
