#!/bin/bash
# resident (1 and 2 groups per lane) vs two-pass RMSD fit on the bench workload (GPU box)
for cfg in "resident=1 resident_groups=1" "resident=1 resident_groups=2" "resident=0"; do
  T=""; for kv in $cfg; do T="$T --tune $kv"; done
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $T > gpurun_out/b_res.json 2> gpurun_out/b_res.err || { tail -5 gpurun_out/b_res.err; exit 1; }
  python - <<PY
import json
j=json.load(open("gpurun_out/b_res.json")); k=j["kernels"]
print("%-36s %9.1f frames/s  ms/step %.3f " % ("$cfg", j["value"], j["ms_per_step"]), {a: b["us_per_frame"] for a, b in k.items() if b["launches"]}, j["roofline"]["frac"])
PY
done
