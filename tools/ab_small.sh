#!/bin/bash
# GPU box: per-kernel trace of the small launches of a step, in-tree library vs ab/libprev.so, three workloads (ONLY=cloth: that one)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for lib in $R/adaptigraph_amd/libadaptigraph_hip.so $R/ab/libprev.so; do
for m in "rope 256 10" "cloth 64 20" "granular 128 10"; do set -- $m
  [ -n "$ONLY" ] && [ "$ONLY" != "$1" ] && continue
  rm -rf /tmp/tr; echo "== $(basename $lib) $1"
  AG_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python $R/bench.py --material $1 --batch $2 --rollout-steps $3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))"
  python $R/tools/rocpd_summary.py trace $(find /tmp/tr -name "*.db" | head -1) | grep -E "bin_kernel|select_lanes|rowptr|edge_node_tab|rollout_step|scan_partial|finalize_connect|fillBuffer" | cut -c1-110
done; done
