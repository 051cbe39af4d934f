#!/bin/bash
# Where a frame's time goes inside a resident launch: a build with -DGR_EXP_TIMELINE=1 (tools/build_variants.sh tl:"-DGR_EXP_TIMELINE=1")
# stamps the device clock at every hand-over of a frame; GR_TIMELINE=1 prints the averages per launch on stderr.
#   tools/timeline_bench.sh <tag> lib.so ...   -> gpurun_out/<tag>_timeline.txt     SIZES: atoms per frame
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}_timeline.txt; : > $OUT
for n in ${SIZES:-520000 1000000}; do
  fps=$(( (768000000 / n + 255) / 256 * 256 ))
  for LIB in "$@"; do
    echo "== $n atoms, $(basename $LIB .so)" >> $OUT
    GR_TIMELINE=1 GR_LIB_PATH=$LIB timeout -k 10 150 python bench.py --atoms $n --steps 6 --warmup 2 --frames-per-step $fps --no-cpu-baseline $TUNES 2>&1 >/dev/null | grep timeline | tail -3 >> $OUT
  done
done
cat $OUT
