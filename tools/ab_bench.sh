#!/bin/bash
# A/B runs of bench.py over several builds of the library (GR_LIB_PATH): tools/ab_bench.sh <tag> lib1.so lib2.so ...
# prints frames/s and the per-kernel microseconds per frame of each build; full lines go to gpurun_out/ab_<tag>_<k>.json
TAG=$1; shift
K=0
for LIB in "$@"; do
  K=$((K+1))
  GR_LIB_PATH=$LIB python bench.py --steps 12 --warmup 3 --no-cpu-baseline $AB_ARGS > gpurun_out/ab_${TAG}_$K.json 2> gpurun_out/ab_${TAG}_$K.err || { echo "$LIB FAILED"; tail -3 gpurun_out/ab_${TAG}_$K.err; continue; }
  python - "$LIB" gpurun_out/ab_${TAG}_$K.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
print("%-40s %9.0f frames/s  sums %.3f  fit %.3f us/frame  path frac %.3f" % (sys.argv[1].split("/")[-1], d["value"], k["k_sums_pk"]["us_per_frame"], k["k_fit_pk"]["us_per_frame"], d["path"]["frac_of_peak"]))
PY
done
