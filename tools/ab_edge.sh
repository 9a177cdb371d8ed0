#!/bin/bash
# GPU box: A/B the edge encoders of precision mode 2: AG_EDGE_PRODUCTS=2 (default: fp16 activations x split-fp16 weights, 3 WG/CU)
# vs 3 (split-bf16: AG_EDGE_ROWS=32 r01 kernel, 33/34/64 the r02 experiments).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
for cfg in "2 32" "3 32" ${AB_EDGE_ALL:+"3 33" "3 34" "3 64"}; do
  set -- $cfg
  AG_EDGE_PRODUCTS=$1 AG_EDGE_ROWS=$2 python tools/time_forward.py 2 20 2>&1 | tail -1
done
for cfg in "2 32" "3 32" ${AB_EDGE_ALL:+"3 33" "3 34" "3 64"}; do
  set -- $cfg
  AG_EDGE_PRODUCTS=$1 AG_EDGE_ROWS=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/ab_bench_$1_$2.json 2> gpurun_out/ab_bench_$1_$2.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_bench_$1_$2.json").read().strip().splitlines()[-1])
print("edge_products=$1 edge_rows=$2", round(d["value"]), "graph-steps/s", {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
PY
done
