#!/bin/bash
# the 510-700 k-atom window with several builds of the library (GR_LIB_PATH): tools/hole_ab.sh <tag> lib.so ... -> gpurun_out/<tag>_hole_ab.txt
# SIZES: atoms per frame (default 520000 600000 1000000); TUNES: extra "--tune k=v" words
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}_hole_ab.txt; : > $OUT
for n in ${SIZES:-520000 600000 1000000}; do
  fps=$(( (768000000 / n + 255) / 256 * 256 ))
  for LIB in "$@"; do
    N=$(basename $LIB .so)
    line=$(GR_LIB_PATH=$LIB timeout -k 10 150 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline $TUNES 2>/dev/null | tail -1) || { echo "$n $N FAILED" >> $OUT; continue; }
    python - "$n" "$N" "$line" >> $OUT <<'PY'
import json, sys
n, label, line = sys.argv[1:4]
d = json.loads(line); r = d['config']['per_rank_resident'][0]
print(f"{int(n):>9} {label:<22} {d['value']:>10.0f} frames/s {1e6 / d['value']:7.3f} us/frame {1e6 / d['value'] / int(n) * 1e6:6.3f} ps/atom  resident launches={r['res_launches']} streams={r['res_last_streams']}")
PY
  done
done
cat $OUT
