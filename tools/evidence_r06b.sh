#!/bin/bash
# GPU box: regenerate round 6 (final sources: agg as q16 rows) committed evidence under gpurun_out/ev6b/ (copy into profiles/ afterwards).
#   tools/evidence_r06.sh [quick]      quick: skip the fuzz / stress / drift / training legs
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; EV=$REPO/gpurun_out/ev6b; rm -rf $EV; mkdir -p $EV
python -c "import torch" 2>/dev/null
python bench.py --steps 20 --warmup 5 > $EV/r06_bench.json 2> $EV/bench.err
python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r06_bench_1stream.json 2>> $EV/bench.err
# BASELINE configs[3] at its global batch on one GPU (8 GPUs would take 64 graphs each)
python bench.py --material cloth --global-batch 512 --rollout-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $EV/r06_bench_cloth512.json 2>> $EV/bench.err
python tools/shared_state_probe.py > $EV/r06_shared_state_probe.txt 2>/dev/null
python tools/fwd_err.py > $EV/r06_fwd_err.txt 2>/dev/null
python tools/dyn_err.py > $EV/r06_dyn_err.txt 2>/dev/null
{
echo "# bench_mpc.py on one MI355X (precision fast), engine option shared_state 1 / 0: BASELINE configs[4] and the reference planner's shipped shape (config/planning/rope.yaml:28-42)"
for ss in 1 0; do
for args in "" "--particles 200 --samples 20000" "--particles 200 --samples 20000 --chunk 500" "--particles 200 --samples 500"; do
  python bench_mpc.py --steps 3 --warmup 1 --shared-state $ss $args 2>> $EV/bench.err | tail -1
done; done
} > $EV/r06_mpc_bench.json
if [ "$1" != "quick" ]; then
python tools/stress_repeat.py 300 > $EV/r06_stress_repeat.txt 2>/dev/null
python tools/fuzz_parity.py 5000 606 > $EV/r06_fuzz_parity.txt 2>/dev/null
rm -f $EV/r06_rollout_drift.txt
AG_DRIFT_FILE=$EV/r06_rollout_drift.txt python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_rollout_at_the_benchmarked_config_vs_oracle" > $EV/drift_pytest.log 2>&1
python bench_train.py --graph > $EV/r06_train_bench.json 2>> $EV/bench.err
fi
cd /tmp && export TMPDIR=/tmp
for m in "rope 256 10" "granular 128 10" "cloth 64 20"; do set -- $m
  rm -rf $EV/trace1
  timeout 300 rocprofv3 --kernel-trace --stats -d $EV/trace1 -o t -- python $REPO/bench.py --material $1 --batch $2 --rollout-steps $3 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r06_bench_1stream_traced_$1.json 2>/dev/null
  python $REPO/tools/rocpd_summary.py trace $(find $EV/trace1 -name "*.db" | head -1) > $EV/r06_kernel_trace_stats_1stream_$1.txt
done
rm -rf $EV/trace1
# the MPPI iteration with the shared-state rollout (and without): where its milliseconds are
for ss in 1 0; do
  rm -rf $EV/trace1
  timeout 300 rocprofv3 --kernel-trace --stats -d $EV/trace1 -o t -- python $REPO/bench_mpc.py --steps 3 --warmup 1 --shared-state $ss > /dev/null 2>&1
  python $REPO/tools/rocpd_summary.py trace $(find $EV/trace1 -name "*.db" | head -1) | head -40 > $EV/r06_mpc_trace_$([ $ss = 1 ] && echo shared || echo plain).txt
done
rm -rf $EV/trace1
# edge encoder / node update / reduce: matrix-pipe busy fraction and clock (SQ counters, their own pass: no trace domains)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $EV/pmc_sq -o pmc -- \
   python $REPO/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-extra > /dev/null 2>&1
python $REPO/tools/rocpd_summary.py pmc $(find $EV/pmc_sq -name "*.db" | head -1) > $EV/r06_sq_pmc.txt 2>/dev/null
rm -rf $EV/pmc_sq
timeout 1500 python $REPO/tools/pmc_traffic.py --out $EV/pmc_traffic.json > /dev/null 2>&1
# the bench line once more, now that the traffic file matches the sources being run (bench.py quotes it only then)
cp $EV/pmc_traffic.json $REPO/profiles/pmc_traffic.json
cd $REPO && python bench.py --steps 20 --warmup 5 > $EV/r06_bench.json 2>> $EV/bench.err
timeout 1200 $REPO/tools/granular_ta.sh > /dev/null 2>&1; cp $REPO/gpurun_out/granular_ta.txt $EV/r06_granular_ta.txt
ls -la $EV
