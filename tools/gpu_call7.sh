#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
CHUNKS="8 16 24 8" timeout -k 10 600 python tools/rmsd_bench.py > gpurun_out/c7_rmsd_bench.json 2> gpurun_out/c7_rmsd_bench.err; echo rc=$?; tail -3 gpurun_out/c7_rmsd_bench.err
python3 -c "
import json
j=json.load(open('gpurun_out/c7_rmsd_bench.json'))
for r in j['results']: print(r['box'][:5], r['pass'][:5], r['chunks'], r['us_per_frame_wall'], r['us_per_frame_kernel'], r['frac_hbm_12B'])
"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tune resident=0 > gpurun_out/c7_bench_twopass.json 2> gpurun_out/c7_bench_twopass.err; echo "bench rc=$?"
python3 -c "
import json
j=json.load(open('gpurun_out/c7_bench_twopass.json')); print(j['value'], j['kernels'])"
timeout -k 10 600 python -m pytest tests/test_gpu_rmsd_fast.py tests/test_gpu_parity.py tests/test_gpu_com_onepass.py -x -q -m gpu 2>&1 | tail -4
