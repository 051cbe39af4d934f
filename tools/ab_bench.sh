#!/bin/bash
# A/B runs of bench.py over several builds of the library (GR_LIB_PATH): tools/ab_bench.sh <tag> lib1.so lib2.so ...
# prints frames/s and the per-kernel microseconds per frame of each build; full lines go to gpurun_out/ab_<tag>_<name>.json
# AB_ARGS: extra bench.py arguments; AB_REPS: runs per build (default 1)
TAG=$1; shift
mkdir -p gpurun_out
for R in $(seq 1 ${AB_REPS:-1}); do
for LIB in "$@"; do
  N=$(basename $LIB .so)
  GR_LIB_PATH=$LIB timeout -k 10 240 python bench.py --steps 12 --warmup 3 --no-cpu-baseline $AB_ARGS > gpurun_out/ab_${TAG}_${N}_$R.json 2> gpurun_out/ab_${TAG}_${N}_$R.err || { echo "$LIB FAILED"; tail -3 gpurun_out/ab_${TAG}_${N}_$R.err; continue; }
  python - "$N" gpurun_out/ab_${TAG}_${N}_$R.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
print("%-28s %9.0f frames/s  resident %.3f  sums %.3f  fit %.3f us/frame" % (sys.argv[1], d["value"], k["k_fit_resident"]["us_per_frame"], k["k_sums_pk"]["us_per_frame"], k["k_fit_pk"]["us_per_frame"]))
PY
done
done
