#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c13_bench.json 2> gpurun_out/c13_bench.err; echo "bench rc=$?"
python3 -c "
import json
j=json.load(open('gpurun_out/c13_bench.json')); print(j['value'], j['roofline']['us_per_frame'], j['roofline']['frac'], j['config']['step_ms'][:5])"
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_resident_fullsize.py tests/test_gpu_resident_fuzz.py -x -q -m gpu > gpurun_out/c13_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c13_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tune resident=0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('two-pass', j['value'], j['kernels']['k_sums_pk'], j['kernels']['k_fit_pk'])"
CHUNKS="8 8" timeout -k 10 600 python tools/rmsd_bench.py > gpurun_out/c13_rmsd_bench.json 2> gpurun_out/c13.err; python3 -c "
import json
j=json.load(open('gpurun_out/c13_rmsd_bench.json'))
for r in j['results']: print(r['box'][:5], r['pass'][:5], r['chunks'], r['us_per_frame_wall'], r['us_per_frame_kernel'], r['frac_hbm_12B'])"
