#!/bin/bash
# GPU box: regenerate the round's committed evidence under gpurun_out/ev/ (copy into profiles/ afterwards).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; EV=$REPO/gpurun_out/ev; rm -rf $EV; mkdir -p $EV
{
echo "# SQ / GRBM counters per launch of the split-bf16 edge encoders at C2 (tools/pmc_fwd.sh: rocprofv3 --pmc, one forward x 6 dispatches);"
echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for shader cycles per launch); SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = pipe-busy cycles;"
echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves."
for spec in "r32_shipped ab/libbase.so AG_EDGE_ROWS=32" "e64_option ab/libbase.so AG_EDGE_ROWS=64" "r32_no_dma_sync_store_split ab/liball.so AG_EDGE_ROWS=32" "e64_no_dma_sync_store_split ab/liball.so AG_EDGE_ROWS=64"; do
  set -- $spec; bash tools/pmc_fwd.sh $1 $2 $3 2>&1 | grep -v "^{" 
done
} > $EV/r02_edge64_pmc.txt 2>&1
{
echo "# tools/trace_e64.py (-DAG_TRACE=1 build), edge_encode_nb_kernel final version: (tag:delta cycles) of wave 0, 6th row tile, blocks 0/1/128/129; each stamp costs ~165 cycles"
AG_EDGE_ROWS=64 AG_LIB_PATH=$REPO/ab/libtrace.so timeout 120 python tools/trace_e64.py 2>/dev/null
} > $EV/r02_edge64_trace.txt
python bench.py --steps 5 --warmup 2 > $EV/r02_bench.json 2> $EV/bench.err
python bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r02_bench_1stream.json 2>> $EV/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $EV/trace1 -o t -- python $REPO/bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $EV/r02_bench_1stream_traced.json 2>/dev/null
python $REPO/tools/rocpd_summary.py trace $(find $EV/trace1 -name "*.db" | head -1) > $EV/r02_kernel_trace_stats_1stream.txt
rm -rf $EV/trace1
timeout 1200 python $REPO/tools/pmc_traffic.py --out $EV/pmc_traffic.json > /dev/null 2>&1
ls -la $EV
