#!/bin/bash
# RMSD-fit of small systems (10 000 - 45 000 atoms): the two-pass path against the resident pass with 16 / 24 / 32 frame streams
# (forced).  -> gpurun_out/small_sweep.txt   (profiles/r03_small_sweep.txt additionally holds the A/B of 8 / 16 / 32 finalizer
# workgroups made with -DGR_RES_MAX_FIN builds, from which the "16 above 8 streams" rule comes)
mkdir -p gpurun_out; OUT=gpurun_out/small_sweep.txt; : > $OUT
for n in ${SIZES:-45000 32817 20000 10000}; do
  fps=$(( (768000000 / n + 255) / 256 * 256 )); [ $fps -gt 32768 ] && fps=32768
  for s in 0 16 24 32; do
    wg=$(( (n + 4095) / 4096 )); [ $s -eq 0 ] || [ $(( wg * s + 2 )) -le 256 ] || continue
    if [ $s -eq 0 ]; then tune="--tune resident=0"; else tune="--tune resident=2 --tune resident_streams=$s"; fi
    line=$(timeout -k 10 200 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline $tune 2>/dev/null | tail -1) || { echo "$n $s FAILED" >> $OUT; exit 1; }
    python -c "
import json,sys
d=json.loads(sys.argv[1]); print('%8d %-10s %10.0f frames/s %.3f us/frame streams=%s' % ($n, 'two-pass' if $s == 0 else 'S=$s', d['value'], 1e6/d['value'], d['config']['per_rank_resident'][0]['res_last_streams']))" "$line" >> $OUT
  done
done
cat $OUT
