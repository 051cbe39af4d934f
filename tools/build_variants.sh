#!/bin/bash
# A/B builds of the library with compile-time switches: tools/build_variants.sh name1:"-DX=1 -DY=0" name2:"..."  -> tools/bin/ab_<name>.so
# (built in parallel; run them on the GPU box with tools/ab_bench.sh <tag> tools/bin/ab_*.so)
cd "$(dirname "$0")/../groan_rs_amd/csrc" || exit 1
mkdir -p ../../tools/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -fno-fast-math -fno-slp-vectorize"
for SPEC in "$@"; do
  NAME=${SPEC%%:*}; DEFS=${SPEC#*:}
  [ "$NAME" = "$SPEC" ] && DEFS=""
  ( /opt/rocm/bin/hipcc $FLAGS $DEFS -o ../../tools/bin/ab_$NAME.so gr_api.hip 2> ../../tools/bin/ab_$NAME.log && echo "built ab_$NAME.so [$DEFS]" || { echo "FAILED ab_$NAME"; tail -5 ../../tools/bin/ab_$NAME.log; } ) &
  # at most 6 compilers at once (8 cores, ~2 GB each)
  while [ "$(jobs -rp | wc -l)" -ge 6 ]; do sleep 1; done
done
wait
