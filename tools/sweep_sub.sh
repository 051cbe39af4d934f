#!/bin/bash
# sub-batch sweep for the RMSD-fit bench (GPU box)
for sb in 128 256 384 768; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --tune sub_batch=$sb > gpurun_out/b.json 2> gpurun_out/b.err || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/b.json")); k=j["kernels"]
print("sb=%-4s %9.1f frames/s  ms/step %.3f  sums %.3f fit %.3f us/frame" % ("$sb", j["value"], j["ms_per_step"], k["k_sums_pk"]["us_per_frame"], k["k_fit_pk"]["us_per_frame"]))
PY
done
