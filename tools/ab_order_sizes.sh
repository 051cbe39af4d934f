#!/bin/bash
# RMSD-fit over frame sizes with the order of a turn left to the library (0), fit first (1), sums first (2): tools/ab_sizes.sh three times on the product library
for order in 0 1 2; do
  EXTRA="--tune resident_fit_last=$order" bash tools/ab_sizes.sh order$order groan_rs_amd/libgroan_hip.so > /dev/null 2>&1
  sed "s/libgroan_hip/fit_last=$order  /" gpurun_out/order${order}_ab_sizes.txt
done | sort -k1,1nr -k2,2
