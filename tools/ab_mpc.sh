#!/bin/bash
# GPU box: bench_mpc.py shapes with one and two rollout streams, for the in-tree library and every ab/lib*.so.  Usage: tools/ab_mpc.sh [reps=2]
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { local tag=$1 lib=$2; shift 2
  AG_LIB_PATH=$lib python bench_mpc.py --steps 3 --warmup 1 "$@" 2>>gpurun_out/ab_mpc_err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('%-34s %9.3f ms  %9.0f graph-steps/s' % ('$tag', d['value'], d['graph_steps_per_s']))"
}
for rep in $(seq ${1:-2}); do
for so in adaptigraph_amd/libadaptigraph_hip.so ab/lib*.so; do
  t=$(basename $so .so); t=${t#lib}
  for s in 1 2; do
    one "$t 1024x15 rope-1k s$s" $PWD/$so --streams $s
    one "$t 20000x15 rope-200 s$s" $PWD/$so --streams $s --particles 200 --samples 20000
    one "$t 500x15 rope-200 s$s" $PWD/$so --streams $s --particles 200 --samples 500
  done
done; done
