#!/bin/bash
# Build timing-only A/B variants of the library: tools/ab_build.sh <tag>=<hipcc -D flags> ...  ->  ab/lib<tag>.so
# e.g. tools/ab_build.sh abl4="-DAG_ABL=4" ord1="-DAG_E64_ORDER=1"
cd "$(dirname "$0")/../adaptigraph_amd/csrc" && mkdir -p ../../ab
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-inline-asm -Wno-unused-value $flags \
     -shared -o ../../ab/lib$tag.so ag_api.hip ag_mlp.hip ag_aggregate.hip ag_edges.hip ag_rollout.hip ag_shared.hip ag_cost.hip ag_train.hip 2>&1 | grep -E "error" &
done
wait
ls -la ../../ab
