#!/bin/bash
# counters of the secondary kernels (centres, translate / wrap, the RMSD passes): PMC passes over tools/secondary_bench.py
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_secondary; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
K=0
for C in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  K=$((K + 1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p$K -- python3 $REPO/tools/secondary_bench.py 1000000 64 > $OUT/run$K.log 2>&1; echo "pass $K rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if not any(x in k[0] for x in ("k_center_sums", "k_sums_pk", "k_translate_wrap", "k_rmsd_accum", "k_fit_pk")): continue
    print(k)
    for c, v in sorted(d.items()): print("    %-24s n=%-4d mean %16.1f" % (c, len(v), sum(v) / len(v)))
PY
