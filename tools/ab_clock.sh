#!/bin/bash
# frames/s, kernel us/frame and the shader clock of the resident launches for several builds (GR_LIB_PATH): tools/ab_clock.sh lib.so ...
for LIB in "$@"; do
  line=$(GR_LIB_PATH=$LIB timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-floor $AB_ARGS 2>/dev/null | tail -1) || { echo "$LIB FAILED"; continue; }
  python3 - "$(basename $LIB .so)" "$line" <<'PY'
import json, sys
j = json.loads(sys.argv[2]); st = j["config"]["per_rank_resident"][0]
print("%-24s %9.0f frames/s  kernel %.3f us/frame  turn %s ns  sclk %s MHz  period %s late %s" % (sys.argv[1], j["value"], j["roofline"]["us_per_frame"], st.get("res_last_turn_ns"), st.get("res_sclk_mhz"), st.get("res_metro_period_ns"), st.get("res_late_permille")))
PY
done
