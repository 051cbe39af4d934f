#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rmsd_fast.py -x -q -m gpu > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/c5_tests.log
timeout -k 10 300 python tools/secondary_bench.py > gpurun_out/c5_secondary.json 2> gpurun_out/c5_secondary.err; echo "secondary rc=$?"
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/c5_secondary.json"))
for b in ("orthorhombic","dodecahedron"):
    for k,v in j[b].items(): print(b[:5], "%-62s"%k, v["us_per_frame"], v["frac_of_hbm_peak"], v["extra_pass_frames_per_call"])
PY
