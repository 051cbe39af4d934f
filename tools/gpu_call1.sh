#!/bin/bash
# round 4, call 1: the proof-failure tests, a default bench line, and ONE --pmc pass at the shape that used to stop in the generator
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py "tests/test_gpu_resident_fullsize.py::test_headline_shape_with_frames_whose_image_proof_fails" -x -q -m gpu > gpurun_out/c1_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/c1_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; echo "bench rc=$?"
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
GROAN_BENCH_TRACE=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/c1_pmc -- python3 $REPO/bench.py --atoms 250000 --steps 2 --warmup 1 --warmup-seconds 0 --frames-per-step 3072 --no-cpu-baseline > $REPO/gpurun_out/c1_pmc.log 2>&1; echo "pmc rc=$?"; tail -5 $REPO/gpurun_out/c1_pmc.log
