#!/bin/bash
# RMSD-fit throughput against system size on the GPU box: the two-pass path beside the resident pass with 1-4 frame streams
# (forced), and what the default tuning chooses.  -> gpurun_out/<tag>_size_sweep.txt
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_size_sweep.txt
mkdir -p gpurun_out; : > $OUT
run() {   # atoms, label, tune args...
    local n=$1 label=$2; shift 2
    local t=""; for kv in "$@"; do t="$t --tune $kv"; done
    local line
    # the same bytes per step and in the pool at every size (768 frames of 1e6 atoms; 3072 of 250 000), 12 + 3 steps
    local fps=$(( (768000000 / n + 255) / 256 * 256 ))
    line=$(timeout -k 10 200 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline $t 2>/dev/null | tail -1) || { echo "$n $label FAILED" >> $OUT; return 1; }
    python - "$n" "$label" "$line" >> $OUT <<'PY'
import json, sys
n, label, line = sys.argv[1:4]
d = json.loads(line)
print(f"{int(n):>9} {label:<22} {d['value']:>12.0f} frames/s  {1e6 / d['value']:7.3f} us/frame  {1e6 / d['value'] / int(n) * 1e6:7.3f} ps/atom  resident={d['config'].get('per_rank_resident')}")
PY
}
for n in ${SIZES:-1000000 800000 700000 500000 400000 330000 250000 200000 125000}; do
    run $n two-pass resident=0 || exit 1
    run $n default || exit 1
    # QUICK=1: only the two passes against the pass with as many streams as fit, whatever part of the chip they fill
    if [ -n "$QUICK" ]; then run $n "any fill" resident=1 resident_fill=1 || exit 1; continue; fi
    for s in ${STREAMS:-1 2 3 4 6 8 12 16}; do
        wg=$(( (n + 4095) / 4096 ))
        if [ $(( wg * s + 2 )) -le 256 ]; then run $n "forced streams=$s" resident=2 resident_streams=$s || exit 1; fi
    done
done
cat $OUT
