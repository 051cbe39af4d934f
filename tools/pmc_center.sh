#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE; KiB units, separate --pmc passes) of the one-pass atoms_center and of the
# one-float4-per-lane translate: tools/pmc_center.sh on the GPU box -> gpurun_out/pmc_cen/summary.txt
R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_cen; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_cen/$C -- python3 $R/tools/pmc_center_run.py > $R/gpurun_out/pmc_cen/$C.log 2>&1; echo "$C rc=$?"
done
python3 - "$R" <<'PY' | tee $R/gpurun_out/pmc_cen/summary.txt
import csv, glob, sys, collections
R=sys.argv[1]
out=collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(R+"/gpurun_out/pmc_cen/%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==C: out[r["Kernel_Name"].split("(")[0]][C].append(float(r["Counter_Value"]))
print("HBM bytes per FRAME of 1e6 atoms (64 frames per launch): FETCH_SIZE x 2 x 1024 / 64, WRITE_SIZE x 1024 / 64; 12 MB = one frame read or written once")
for k,v in sorted(out.items()):
    if "resident" in k or "translate" in k or "center_sums" in k:
        f=v.get("FETCH_SIZE",[0.0]); w=v.get("WRITE_SIZE",[0.0])
        print("%-64s launches %3d  read %.2f MB  written %.2f MB" % (k[:64], len(f), 2*1024*sum(f)/len(f)/64/1e6, 1024*sum(w)/max(len(w),1)/64/1e6))
PY
