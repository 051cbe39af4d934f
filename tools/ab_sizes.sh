#!/bin/bash
# default tuning over several frame sizes for several builds (GR_LIB_PATH): tools/ab_sizes.sh <tag> lib.so ...   SIZES="..."  EXTRA="--tune k=v"
TAG=$1; shift
OUT=gpurun_out/${TAG}_ab_sizes.txt; mkdir -p gpurun_out; : > $OUT
for n in ${SIZES:-1000000 800000 600000 520000 330000 250000 125000 62000}; do
  fps=$(( (768000000 / n + 255) / 256 * 256 ))
  for LIB in "$@"; do
    line=$(GR_LIB_PATH=$LIB timeout -k 10 200 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline --no-live-floor $EXTRA 2>/dev/null | tail -1) || { echo "$n $(basename $LIB .so) FAILED" >> $OUT; continue; }
    python3 - "$n" "$(basename $LIB .so)" "$line" >> $OUT <<'PY'
import json, sys
n, label, line = sys.argv[1:4]
d = json.loads(line); st = d["config"]["per_rank_resident"][0]
print(f"{int(n):>9} {label:<18} {d['value']:>12.0f} frames/s  {1e6 / d['value']:7.3f} us/frame  streams={st.get('res_last_streams')} launches={st.get('res_launches')} period={st.get('res_metro_period_ns')} turn={st.get('res_last_turn_ns')} late={st.get('res_late_permille')}")
PY
  done
done
cat $OUT
