#!/bin/bash
# small systems: more streams / more finalizers
mkdir -p gpurun_out; OUT=gpurun_out/small_sweep.txt; : > $OUT
for n in 45000 32817 20000 10000; do
  fps=$(( (768000000 / n + 255) / 256 * 256 )); [ $fps -gt 32768 ] && fps=32768
  for lib in s32 s32f16 s32f32; do
    for s in 16 24 32; do
      wg=$(( (n + 4095) / 4096 )); [ $(( wg * s + 2 )) -le 256 ] || continue
      line=$(GR_LIB_PATH=tools/bin/ab_$lib.so timeout -k 10 200 python bench.py --atoms $n --steps 12 --warmup 3 --frames-per-step $fps --no-cpu-baseline --tune resident=2 --tune resident_streams=$s 2>/dev/null | tail -1) || { echo "$n $lib $s FAILED" >> $OUT; exit 1; }
      python -c "
import json,sys
d=json.loads(sys.argv[1]); print('%8d %-8s S=%-3d %10.0f frames/s %.3f us/frame streams=%s' % ($n, '$lib', $s, d['value'], 1e6/d['value'], d['config']['per_rank_resident'][0]['res_last_streams']))" "$line" >> $OUT
    done
  done
done
cat $OUT
