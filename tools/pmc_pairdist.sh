#!/bin/bash
# instructions per pair of k_pairdist (GPU box): one PMC pass over tools/pairdist_bench.py
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_pairdist; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $REPO/tools/pairdist_bench.py > $OUT/run.log 2>&1; echo "rc=$?"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pairdist" in r["Kernel_Name"]: agg[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
