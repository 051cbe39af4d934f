#!/bin/bash
# frames parked on chip (GR_RES_K) against the turn time: tools/build_variants.sh k6:"" k5:"-DGR_RES_K=5" k4:"-DGR_RES_K=4", then this on the GPU box
for L in k6 k5 k4 k6; do
  echo "== $L"
  AB_ARGS="--tune resident_metro_ns=1" bash tools/ab_clock.sh tools/bin/ab_$L.so
  GR_LIB_PATH=tools/bin/ab_$L.so timeout -k 10 200 python tools/center_metro.py 2>&1 | grep "metronome 1 ns" | head -2
done
