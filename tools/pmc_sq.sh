#!/bin/bash
# SQ counters per kernel for the bench command (run on the GPU box via gpurun); prints per-kernel averages.
set -o pipefail
ARGS=${1:-"--steps 3 --warmup 1 --no-cpu-baseline"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $REPO/bench.py $ARGS > $OUT/$N.log 2>&1
  echo "pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"], k)
        if key not in seen: seen.add(key); cnt[(f, k)] += 1
disp = collections.defaultdict(dict)
for (f, k), n in cnt.items(): disp[k][f] = n
for k, d in agg.items():
    n = max(disp[k].values())
    print(k, "dispatches", n)
    for c, v in sorted(d.items()): print("   %-24s %16.1f per dispatch" % (c, v / n))
PY
