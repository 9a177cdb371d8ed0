#!/bin/bash
# GPU box: regenerate round 3's committed evidence under gpurun_out/ev/ (copy into profiles/ afterwards).
# Optional: ab/libexp.so (tools/ab_build.sh exp="-DAG_EXPERIMENTS") for the segment-reduce experiments' PMC table.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; EV=$REPO/gpurun_out/ev; rm -rf $EV; mkdir -p $EV
python -c "import torch" 2>/dev/null
python bench.py --steps 20 --warmup 5 > $EV/r03_bench.json 2> $EV/bench.err
python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r03_bench_1stream.json 2>> $EV/bench.err
python bench.py --steps 5 --warmup 2 --weights trained_rope --no-cpu-baseline --no-extra > $EV/r03_bench_trained_weights.json 2>> $EV/bench.err
AG_NODE_DEDUP=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $EV/r03_bench_no_dedup.json 2>> $EV/bench.err
{
echo "# bench_mpc.py on one MI355X (precision fast): BASELINE configs[4] and the reference planner's shipped shape (config/planning/rope.yaml:28-42)"
for args in "" "--particles 200 --samples 20000" "--particles 200 --samples 20000 --chunk 500" "--particles 200 --samples 500" "--particles 100 --samples 500"; do
  python bench_mpc.py --steps 3 --warmup 1 $args 2>> $EV/bench.err | tail -1
done
} > $EV/r03_mpc_bench.json
python bench_train.py --graph > $EV/r03_train_bench.json 2>> $EV/bench.err
{
echo "# tools/batch_sweep.sh on one MI355X (r03 kernels, precision fast, 2 rollout streams): per-GPU shares of the strong-scaling configs"
echo "cloth-4k, 20-step rollout"; bash tools/batch_sweep.sh cloth 20 "64 128 256 512"
echo "rope-1k, 10-step rollout"; bash tools/batch_sweep.sh rope 10 "32 64 128 256 512 1024"
echo "granular-2k, 10-step rollout"; bash tools/batch_sweep.sh granular 10 "16 32 64 128"
} > $EV/r03_batch_sweep.txt 2>> $EV/bench.err
python tools/fwd_err.py > $EV/r03_fwd_err.txt 2>/dev/null
python tools/dyn_err.py > $EV/r03_dyn_err.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for m in "rope 256 10" "granular 128 10" "cloth 64 20"; do set -- $m
  rm -rf $EV/trace1
  timeout 300 rocprofv3 --kernel-trace --stats -d $EV/trace1 -o t -- python $REPO/bench.py --material $1 --batch $2 --rollout-steps $3 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r03_bench_1stream_traced_$1.json 2>/dev/null
  python $REPO/tools/rocpd_summary.py trace $(find $EV/trace1 -name "*.db" | head -1) > $EV/r03_kernel_trace_stats_1stream_$1.txt
done
rm -rf $EV/trace1
timeout 1500 python $REPO/tools/pmc_traffic.py --out $EV/pmc_traffic.json > /dev/null 2>&1
if [ -f $REPO/ab/libexp.so ]; then
  bash $REPO/tools/agg_pmc.sh $REPO/ab/libexp.so > /dev/null 2>&1; cp $REPO/gpurun_out/agg_pmc.txt $EV/r03_agg_pmc.txt
  AG_LIB_PATH=$REPO/ab/libexp.so python $REPO/tools/agg_check.py rope 1000 256 20 > $EV/r03_agg_stream_vs_half.txt 2>&1
fi
ls -la $EV
