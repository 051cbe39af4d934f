#!/bin/bash
# sweep sub-batch x accumulate chunks x fit workgroups for the RMSD-fit bench (run on the GPU box); prints one line per point
for sb in ${SBS:-32 48 64}; do for ch in ${CHS:-24 32 48}; do for fw in ${FWS:-64 128 256}; do
  GR_SUB_BATCH=$sb GR_CHUNKS=$ch GR_FIT_WGS=$fw GR_OVERLAP=${OV:-0} timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-8} > gpurun_out/b.json 2> gpurun_out/b.err
  python - <<PY
import json
j=json.load(open("gpurun_out/b.json")); k=j["kernels"]
print("sb=%-3s ch=%-3s fw=%-3s  %9.1f frames/s  acc %.3f fin %.3f fit %.3f" % ("$sb","$ch","$fw", j["value"], k["k_rmsd_accum"]["us_per_frame"], k["k_rmsd_finalize"]["us_per_frame"], k["k_fit"]["us_per_frame"]))
PY
done; done; done
