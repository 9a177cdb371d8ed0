#!/bin/bash
# GPU box: socket power and shader clock (rocm-smi, sampled once a second) under (a) pure bf16 MFMA chains, constant and random
# operands, (b) the engine's forward loop in the default and the exact-fp32 mode.  -> gpurun_out/power_probe.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO && mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
: > $OUT
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_power tools/ubench/mfma_power.hip 2>>$OUT || exit 1
sample() {   # $1 = pid to watch
  while kill -0 $1 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' ';' >> $OUT
    echo >> $OUT
    sleep 1
  done
}
python -c "import torch" 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> $OUT
for cfg in "0 2 5 0" "1 2 5 0" "1 1 5 0" "1 2 5 1"; do
  echo "== mfma_power $cfg" >> $OUT
  /tmp/mfma_power $cfg >> $OUT 2>&1 &
  pid=$!
  sleep 1.5; sample $pid; wait $pid
done
for prec in 2 0; do
  echo "== forward loop, precision $prec" >> $OUT
  python tools/time_forward.py $prec 2500 >> $OUT 2>&1 &
  pid=$!
  sleep 6; sample $pid; wait $pid
done
cat $OUT
