#!/bin/bash
# counters of the pair-distance kernels (GPU box): PMC passes over tools/pairdist_bench.py, averages per kernel and grid
# PD_ONLY=triclinic limits the run to one cell; PMC_SETS="a b c|d e f" overrides the counter sets (one pass each)
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_pairdist; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
SETS=${PMC_SETS:-"SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES|SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"}
IFS='|' read -ra LIST <<< "$SETS"
K=0
for C in "${LIST[@]}"; do
  K=$((K + 1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$K -- python3 $REPO/tools/pairdist_bench.py > $OUT/run$K.log 2>&1; echo "pass $K rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pairdist" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()): print("    %-24s n=%-4d mean %16.1f   (first half %16.1f, second half %16.1f)" % (c, len(v), sum(v) / len(v), sum(v[:len(v)//2]) / max(1, len(v)//2), sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2)))
PY
