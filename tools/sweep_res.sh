#!/bin/bash
# resident vs two-pass RMSD fit on the bench workload (GPU box)
for r in 1 0; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --tune resident=$r > gpurun_out/b_res$r.json 2> gpurun_out/b_res$r.err || { tail -5 gpurun_out/b_res$r.err; exit 1; }
  python - <<PY
import json
j=json.load(open("gpurun_out/b_res$r.json")); k=j["kernels"]
print("resident=$r %9.1f frames/s  ms/step %.3f  frames/step %d " % (j["value"], j["ms_per_step"], j["config"]["frames_per_step"]), {a: b["us_per_frame"] for a, b in k.items()}, j["roofline"]["kernel"], j["roofline"]["frac"], j["path"])
PY
done
