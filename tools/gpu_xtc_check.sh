#!/bin/bash
# GPU box: xtc device tests, stage rates with the host-phase trace, and a kernel trace of the same tool (-> gpurun_out/)
set -o pipefail
mkdir -p gpurun_out && export TMPDIR=/tmp
R=$(pwd)
timeout -k 10 500 python -m pytest tests/test_gpu_xtc_device.py tests/test_gpu_xtc_pipeline.py tests/test_gpu_xtc_writer.py -x -q > gpurun_out/xtc_tests.log 2>&1; tail -3 gpurun_out/xtc_tests.log
GR_XTC_TRACE=1 timeout -k 10 300 python tools/xtc_bench.py > gpurun_out/xtc_bench.json 2> gpurun_out/xtc_bench.err && cat gpurun_out/xtc_bench.json &&
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xtcprof -o xtc -- python3 $R/tools/xtc_bench.py --frames 64 > /dev/null 2>&1; find $R/gpurun_out/xtcprof -name "*kernel_stats.csv" -exec head -8 {} ;
