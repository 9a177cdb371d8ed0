#!/bin/bash
# GPU box: regenerate the round's committed evidence under gpurun_out/ev/ (copy into profiles/ afterwards).
# Needs ab/libtrace.so (-DAG_TRACE=1) and the streaming-kernel ablation builds ab/lib{base,nodma,nobar,nostore,fake,all}.so
# (tools/ab_build.sh base="" nodma="-DAG_ABL=4" nobar="-DAG_ABL=18" nostore="-DAG_ABL=8" fake="-DAG_ABL=64" all="-DAG_ABL=94" trace="-DAG_TRACE=1").
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; EV=$REPO/gpurun_out/ev; rm -rf $EV; mkdir -p $EV
python -c "import torch" 2>/dev/null
{
echo "# SQ / GRBM counters per launch of the edge encoders at C2 (tools/pmc_fwd.sh: rocprofv3 --pmc, one forward x 6 dispatches), precision mode 2;"
echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for shader cycles per launch); SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = pipe-busy cycles;"
echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves.  First line of each block: HIP-event ms per launch (same run)."
for spec in "weight_stationary_default default AG_EDGE_WS=1" "streaming_two_product default AG_EDGE_WS=0" "streaming_split_bf16 default AG_EDGE_PRODUCTS=3"; do
  set -- $spec; bash tools/pmc_fwd.sh $1 $2 $3 2>&1 | grep -v "^$" | sed -n '1,12p'
done
} > $EV/r02_edge_pmc.txt 2>&1
{
echo "# tools/trace_ws.py (-DAG_TRACE=1 build): weight-stationary edge encoder, workgroup 3, rounds 100..107, s_memtime deltas per wave:"
echo "# | @round start  2:+first phase  3:+second phase (arrives at the barrier)  4:+barrier passed.   Each stamp costs ~100 cycles."
AG_LIB_PATH=$REPO/ab/libtrace.so timeout 120 python tools/trace_ws.py 2>/dev/null
} > $EV/r02_edge_ws_trace.txt
{
echo "# Timing-only ablations of the STREAMING two-product edge kernel (AG_EDGE_WS=0; tools/ab_run.sh, ms per launch at C2):"
echo "# base | fake = synthetic edge indices | nobar = no barrier / DMA drain | nodma = no L2->LDS weight copies | nostore = no Eterm store | all"
AG_EDGE_WS=0 bash tools/ab_run.sh 2 20 2>&1 | grep -v trace
} > $EV/r02_edge_stream_ablation.txt
{
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/lone tools/ubench/mfma_lone.hip && /tmp/lone
hipcc --offload-arch=gfx950 -O3 -o /tmp/den tools/ubench/mfma_f16_denorm.hip && /tmp/den
} > $EV/r02_mfma_ubench.txt 2>&1
timeout 300 bash tools/power_probe.sh > /dev/null 2>&1; cp gpurun_out/power_probe.txt $EV/r02_power_probe.txt
timeout 200 python tools/fwd_err.py > $EV/r02_fwd_err.txt 2>/dev/null
python bench.py --steps 5 --warmup 2 > $EV/r02_bench.json 2> $EV/bench.err
python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r02_bench_1stream.json 2>> $EV/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $EV/trace1 -o t -- python $REPO/bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r02_bench_1stream_traced.json 2>/dev/null
python $REPO/tools/rocpd_summary.py trace $(find $EV/trace1 -name "*.db" | head -1) > $EV/r02_kernel_trace_stats_1stream.txt
rm -rf $EV/trace1
timeout 1200 python $REPO/tools/pmc_traffic.py --out $EV/pmc_traffic.json > /dev/null 2>&1
ls -la $EV
