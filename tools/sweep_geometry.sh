#!/bin/bash
# sweep sub-batch x sums chunks x fit workgroups for the RMSD-fit bench (run on the GPU box); prints one line per point
for sb in ${SBS:-32 48 64}; do for ch in ${CHS:-24 32 48}; do for fw in ${FWS:-64 128 256}; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-8} --tune sub_batch=$sb --tune chunks=$ch --tune fit_wgs=$fw > gpurun_out/b.json 2> gpurun_out/b.err || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/b.json")); k=j["kernels"]
print("sb=%-3s ch=%-3s fw=%-3s  %9.1f frames/s  sums %.3f fit %.3f us/frame" % ("$sb","$ch","$fw", j["value"], k["k_sums_pk"]["us_per_frame"], k["k_fit_pk"]["us_per_frame"]))
PY
done; done; done
