#!/bin/bash
# GPU box: A/B the edge encoders of precision mode 2: AG_EDGE_WS=1 weight-stationary kernel vs 0 streaming two-product kernel
# (AG_EDGE_PRODUCTS=2); AG_EDGE_PRODUCTS=3 = split-bf16 (AG_EDGE_ROWS=32 r01 kernel; AB_EDGE_ALL=1 adds the 33/34/64 experiments).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
CFGS=("2 32 1" "2 32 0" "3 32 0")
[ -n "$AB_EDGE_ALL" ] && CFGS+=("3 33 0" "3 34 0" "3 64 0")
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  AG_EDGE_PRODUCTS=$1 AG_EDGE_ROWS=$2 AG_EDGE_WS=$3 python tools/time_forward.py 2 20 2>&1 | tail -1
done
for rep in 1 2; do
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  AG_EDGE_PRODUCTS=$1 AG_EDGE_ROWS=$2 AG_EDGE_WS=$3 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/ab_bench_$1_$2_$3.json 2> gpurun_out/ab_bench_$1_$2_$3.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_bench_$1_$2_$3.json").read().strip().splitlines()[-1])
print("edge_products=$1 edge_rows=$2 stationary=$3", round(d["value"]), "graph-steps/s", {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
PY
done
done
