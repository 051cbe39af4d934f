#!/bin/bash
# the resident pass's metronome (gr_resident.h) on the benchmark itself: off, fixed periods, the controller.  GPU box.
# usage: tools/metro_sweep.sh <tag> [atoms] [periods...]
TAG=${1:-r05}; N=${2:-1000000}; shift 2
P=${@:-"1 4200 4000 3900 3800 3700 0"}
OUT=gpurun_out/${TAG}_metro_sweep.txt
mkdir -p gpurun_out; : > $OUT
for T in $P; do
  line=$(timeout -k 10 200 python bench.py --atoms $N --steps 10 --warmup 3 --no-cpu-baseline --no-live-floor --tune resident_metro_ns=$T 2>/dev/null | tail -1) || { echo "T=$T FAILED" >> $OUT; continue; }
  python3 - "$T" "$line" >> $OUT <<'PY'
import json, sys
T, j = sys.argv[1], json.loads(sys.argv[2])
r = j["roofline"]; st = j["config"]["per_rank_resident"][0]
print("metro_ns=%-6s %9.1f frames/s  kernel %.4f us/frame  frac %.4f  period %s ns  last turn %s ns  late %s permille  sclk %s MHz  step_ms %s" % (
    T, j["value"], r["us_per_frame"], r["frac"], st.get("res_metro_period_ns"), st.get("res_last_turn_ns"), st.get("res_late_permille"), st.get("res_sclk_mhz"), j["config"]["step_ms"][-4:]))
PY
done
cat $OUT
