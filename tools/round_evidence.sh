#!/bin/bash
# Round evidence on the GPU box (-> gpurun_out/): kernel trace + PMC passes of the bench command, the memory floors, the secondary
# pipelines, pair distances, the RMSD-without-fit pass, config 5 at length.  Copy what is to be judged into profiles/ afterwards.
#   bash tools/round_evidence.sh r04 [part ...]     parts: a (secondary, ceilings) b (bench profile + PMC) c (streams profile, pairdist)
#                                                          d (sweeps, two-pass bench, fuzz) e (rmsd pass profile, config 5, bench line)
set -o pipefail
TAG=${1:-r04}; shift
PARTS=${*:-a b c d e}
mkdir -p gpurun_out
BENCH_PROF="--steps 5 --warmup 2 --warmup-seconds 0 --frames-per-step 768 --no-cpu-baseline"
for P in $PARTS; do case $P in
a)  # (the secondary pipelines first: freeing the gigabytes of an earlier section perturbs the next timed region for a while)
    timeout -k 10 300 python tools/secondary_bench.py > gpurun_out/${TAG}_secondary.json 2> gpurun_out/${TAG}_secondary.err; echo "secondary rc=$?"
    sleep 3
    timeout -k 10 200 tools/bin/ceiling_bench > gpurun_out/${TAG}_ceiling_grid.json 2> gpurun_out/${TAG}_ceiling.err; echo "ceiling (grid) rc=$?"
    timeout -k 10 200 tools/bin/ceiling_resident > gpurun_out/${TAG}_ceiling_resident.json 2>> gpurun_out/${TAG}_ceiling.err; echo "ceiling (persistent) rc=$?"
    timeout -k 10 300 tools/bin/store_variants > gpurun_out/${TAG}_store_variants.json 2>> gpurun_out/${TAG}_ceiling.err; echo "store variants rc=$?"
    CHUNKS="0 8 16" timeout -k 10 600 python tools/rmsd_bench.py > gpurun_out/${TAG}_rmsd_bench.json 2> gpurun_out/${TAG}_rmsd_bench.err; echo "rmsd bench rc=$?" ;;
b)  PROFILE_ARGS="$BENCH_PROF" timeout -k 10 1000 bash tools/profile.sh $TAG "$BENCH_PROF" > gpurun_out/${TAG}_profile.log 2>&1; echo "profile rc=$?"; tail -4 gpurun_out/${TAG}_profile.log ;;
c)  # frame streams AT THE SHAPE THAT USED TO STOP under --pmc (rounds 1-3: 9216 one-box copies queued ahead of the generator kernels)
    S="--atoms 250000 --steps 3 --warmup 1 --warmup-seconds 0 --frames-per-step 3072 --no-cpu-baseline"
    PROFILE_ARGS="$S" PASS_TIMEOUT=240 timeout -k 10 1000 bash tools/profile.sh ${TAG}_streams "$S" > gpurun_out/${TAG}_streams_profile.log 2>&1; echo "streams profile rc=$?"; tail -3 gpurun_out/${TAG}_streams_profile.log
    timeout -k 10 300 python tools/pairdist_bench.py > gpurun_out/${TAG}_pairdist.json 2> gpurun_out/${TAG}_pairdist.err; echo "pairdist rc=$?"
    PD_ONLY=triclinic timeout -k 10 400 bash tools/pmc_pairdist.sh > gpurun_out/${TAG}_pmc_pairdist_sym.txt 2>&1; echo "pmc pairdist rc=$?" ;;
d)  QUICK=1 SIZES="1000000 900000 800000 700000 650000 600000 550000 520000 500000 420000 330000 290000 250000 200000 160000 125000 100000 80000 62000 45000 32817 20000" timeout -k 10 1150 bash tools/size_sweep.sh ${TAG} > gpurun_out/${TAG}_size_sweep.log 2>&1; echo "size sweep rc=$?" ;;
d2) timeout -k 10 900 bash tools/hole_sweep.sh ${TAG} > gpurun_out/${TAG}_hole_sweep.log 2>&1; echo "hole sweep rc=$?"
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune resident=0 > gpurun_out/${TAG}_bench_twopass.json 2> gpurun_out/${TAG}_bench_twopass.err; echo "two-pass bench rc=$?"
    timeout -k 10 400 python tools/resident_fuzz.py 240 4 > gpurun_out/${TAG}_resident_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/${TAG}_resident_fuzz.txt ;;
e)  # the RMSD-without-fit pass under the profiler (kernel trace, then the HBM counters), config 5 at length, the bench line with its CPU leg
    export TMPDIR=/tmp; REPO=$(pwd)
    ( cd /tmp; CHUNKS="0" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_rmsd/trace -- python3 $REPO/tools/rmsd_bench.py > $REPO/gpurun_out/prof_${TAG}_rmsd_trace.log 2>&1; echo "rmsd trace rc=$?"
      for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
        N=$(echo $C | tr ' ' '_' | cut -c1-24)
        CHUNKS="0" timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/prof_${TAG}_rmsd/pmc_$N -- python3 $REPO/tools/rmsd_bench.py > $REPO/gpurun_out/prof_${TAG}_rmsd_pmc_$N.log 2>&1; echo "rmsd pmc $N rc=$?"
      done )
    python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_rmsd gpurun_out/${TAG}_rmsd
    timeout -k 10 900 python tools/c5_pipeline.py > gpurun_out/${TAG}_xtc_pipeline.json 2> gpurun_out/${TAG}_xtc_pipeline.err; echo "c5 rc=$?"
    timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" ;;
esac; done
