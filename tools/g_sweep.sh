#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/g_sweep.txt; : > $OUT
run() { n=$1; s=$2; g=$3; fps=$(( (768000000 / n + 255) / 256 * 256 ));
  line=$(timeout -k 10 150 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline --tune resident=2 --tune resident_streams=$s --tune resident_wg_groups=$g 2>/dev/null | tail -1) || { echo "$n S=$s G=$g FAILED" >> $OUT; return; }
  python -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['config']['per_rank_resident'][0]; print('%8d S=%d G=%4d %10.0f frames/s %.3f us/frame  launches=%s streams=%s' % ($n, $s, $g, d['value'], 1e6/d['value'], r['res_launches'], r['res_last_streams']))" "$line" >> $OUT; }
for g in 1024 896 768 704 640; do run 600000 1 $g; done
for g in 1024 832 768 704; do run 700000 1 $g; done
for g in 1024 768 640; do run 380000 2 $g; done
for g in 1024 768 512; do run 1000000 1 $g; done
cat $OUT
