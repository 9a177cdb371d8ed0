#!/bin/bash
# GPU box: streamed vs per-node segment reduce for every ab/lib*.so variant (tools/agg_check.py), then the other materials on the default build.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
{
echo "== default build"; timeout 300 python tools/agg_check.py rope 1000 256 20 2>&1 | tail -3
for so in ab/lib*.so; do
  echo "== $so"; AG_LIB_PATH=$REPO/$so timeout 300 python tools/agg_check.py rope 1000 256 20 2>&1 | tail -3
done
echo "== granular / cloth / small, default build"
timeout 300 python tools/agg_check.py granular 2000 128 10 2>&1 | tail -3
timeout 300 python tools/agg_check.py cloth 4096 64 10 2>&1 | tail -3
timeout 300 python tools/agg_check.py rope 200 500 20 2>&1 | tail -3
timeout 300 python tools/agg_check.py rope 63 3 5 2>&1 | tail -3
} 2>&1 | tee gpurun_out/agg_ab.txt
