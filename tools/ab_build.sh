#!/bin/bash
# Build timing-only ablation variants of the library: tools/ab_build.sh <AG_ABL value> ...  ->  ab/libabl_<v>.so
cd "$(dirname "$0")/../adaptigraph_amd/csrc" && mkdir -p ../../ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-inline-asm -Wno-unused-value -DAG_ABL=$v \
     -shared -o ../../ab/libabl_$v.so ag_api.hip ag_mlp.hip ag_aggregate.hip ag_edges.hip ag_rollout.hip ag_cost.hip ag_train.hip 2>&1 | grep -E "error" &
done
wait
ls -la ../../ab
