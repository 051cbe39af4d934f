#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box via gpurun):
#   1. kernel trace + stats  2..n. separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters)
# Summaries land in gpurun_out/prof_<tag>/; copy the ones to be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
# the bench with its OWN warm-up (0.5 s of launches on the warm-up slots): the averages of the trace are then steady-state figures
# (r04 profiled --warmup-seconds 0, i.e. the launches the bench itself discards, and read 5.7 % slow)
ARGS=${2:-"--steps 20 --warmup 3 --warmup-seconds 0.5 --no-cpu-baseline --no-live-floor"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" ; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  # (a pass that hangs -- seen with rocprofv3 --pmc around the frame generator's largest launches -- must not take the others with it)
  timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?"
done
PROFILE_ARGS="$ARGS" python3 $REPO/tools/pmc_summary.py $OUT $OUT/$TAG
ls $OUT
