#!/bin/bash
# Round evidence on the GPU box (-> gpurun_out/): kernel trace + PMC passes of the bench command, the memory floors, the secondary
# pipelines, pair distances.  Copy what is to be judged into profiles/ afterwards (tools/collect_evidence.sh).
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out
# (the secondary pipelines first: freeing the gigabytes of an earlier section perturbs the next timed region for a while)
timeout -k 10 300 python tools/secondary_bench.py > gpurun_out/${TAG}_secondary.json 2> gpurun_out/${TAG}_secondary.err; echo "secondary rc=$?"
sleep 3
timeout -k 10 200 tools/bin/ceiling_bench > gpurun_out/${TAG}_ceiling.json 2> gpurun_out/${TAG}_ceiling.err; echo "ceiling rc=$?"
PROFILE_ARGS="--steps 5 --warmup 2 --frames-per-step 768 --no-cpu-baseline" timeout -k 10 900 bash tools/profile.sh $TAG "--steps 5 --warmup 2 --frames-per-step 768 --no-cpu-baseline" > gpurun_out/${TAG}_profile.log 2>&1; echo "profile rc=$?"; tail -4 gpurun_out/${TAG}_profile.log
timeout -k 10 300 python tools/pairdist_bench.py > gpurun_out/${TAG}_pairdist.json 2> gpurun_out/${TAG}_pairdist.err; echo "pairdist rc=$?"
PD_ONLY=triclinic timeout -k 10 300 bash tools/pmc_pairdist.sh > gpurun_out/${TAG}_pmc_pairdist_sym.txt 2>&1; echo "pmc pairdist rc=$?"
# frame streams: the same evidence for launches of 4 streams of 250 000 atoms, 1024 frames each (with 3072 frames per step the
# frame generator's 9216-frame launch hung under rocprofv3 --pmc every other time; GROAN_BENCH_TRACE=1 shows where a run stops)
PASS_TIMEOUT=120 timeout -k 10 900 bash tools/profile.sh ${TAG}_streams "--atoms 250000 --steps 3 --warmup 1 --frames-per-step 1024 --no-cpu-baseline" > gpurun_out/${TAG}_streams_profile.log 2>&1; echo "streams profile rc=$?"; tail -3 gpurun_out/${TAG}_streams_profile.log
# sizes: two-pass path / default / the pass with as many streams as fit whatever they fill
QUICK=1 SIZES="1000000 900000 800000 700000 600000 500000 420000 330000 290000 250000 200000 160000 125000 100000 80000 62000 45000 32817 20000" timeout -k 10 1100 bash tools/size_sweep.sh ${TAG} > gpurun_out/${TAG}_size_sweep.log 2>&1; echo "size sweep rc=$?"
# the two-pass path (what frames that do not fill the chip, sub-selections that are not the whole system ... take): resident pass off
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune resident=0 > gpurun_out/${TAG}_bench_twopass.json 2> gpurun_out/${TAG}_bench_twopass.err; echo "two-pass bench rc=$?"
