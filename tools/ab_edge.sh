#!/bin/bash
# GPU box: A/B the two split-bf16 edge encoders (AG_EDGE_ROWS=32: r01 kernel, 64: two row blocks per wave).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
for r in 32 33 64; do
  AG_EDGE_ROWS=$r python tools/time_forward.py 2 20 2>&1 | tail -1
  AG_EDGE_ROWS=$r python tools/time_forward.py 1 20 2>&1 | tail -1
done
for r in 32 33 64; do
  AG_EDGE_ROWS=$r python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/ab_bench_$r.json 2> gpurun_out/ab_bench_$r.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_bench_$r.json").read().strip().splitlines()[-1])
print("edge_rows=$r", round(d["value"]), "graph-steps/s", {k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
PY
done
