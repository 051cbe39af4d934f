#!/bin/bash
# the 510-690 k-atom window: one frame fills 0.5-0.67 of the chip, two do not fit.  Two-pass path, default, and
# the resident pass with ONE stream cut into workgroups of G groups (GR_TUNE_RESIDENT_WG_GROUPS) -> gpurun_out/<tag>_hole_sweep.txt
mkdir -p gpurun_out; OUT=gpurun_out/${1:-r04}_hole_sweep.txt; : > $OUT
run() { n=$1; label=$2; shift 2; t=""; for kv in "$@"; do t="$t --tune $kv"; done
  fps=$(( (768000000 / n + 255) / 256 * 256 ))
  line=$(timeout -k 10 150 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline $t 2>/dev/null | tail -1) || { echo "$n $label FAILED" >> $OUT; return; }
  python - "$n" "$label" "$line" >> $OUT <<'PY'
import json, sys
n, label, line = sys.argv[1:4]
d = json.loads(line); r = d['config']['per_rank_resident'][0]
print(f"{int(n):>9} {label:<14} {d['value']:>10.0f} frames/s {1e6 / d['value']:7.3f} us/frame {1e6 / d['value'] / int(n) * 1e6:6.3f} ps/atom  resident launches={r['res_launches']} streams={r['res_last_streams']}")
PY
}
for n in ${SIZES:-520000 560000 600000 650000 690000}; do
  run $n two-pass resident=0
  run $n default
  for g in ${GS:-1024 832 768 704 640 576 512}; do
    wg=$(( ((n + 255) / 256 * 64 + g - 1) / g ))
    if [ $(( wg + 2 )) -le 254 ]; then run $n "G=$g($wg wg)" resident=2 resident_streams=1 resident_wg_groups=$g; fi
  done
done
cat $OUT
