cd $GRAFT_REPO_ROOT
timeout 300 python tools/ws_check.py 2>&1 | grep -v amdgpu.ids
for v in 1 2; do
  AG_EDGE_WS=$v timeout 300 python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('AG_EDGE_WS=$v 1 stream', round(d['value']), 'edge ms', d['roofline']['avg_launch_ms'], 'status', d['config']['model_status'])"
done
python -m pytest tests/test_gpu_parity.py -x -q -k "golden or stationary" 2>&1 | tail -3
