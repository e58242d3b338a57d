#!/bin/bash
# One round's measurements of record, in stages short enough for a gpurun call each:  tools/collect_round.sh r04 STAGE
#   a: c3 (counters) + the default bench line      b: c2, c1, the MansionLR-shaped dense pile step (100 views, and 151: an on-chip rung)
#   c: c5 slice + c5 full size + its e2e line       d: rows around the path (c2), published-shape context lines
#   e: fuzz campaigns on this tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
rnd=$1; stage=$2
cd $R
case $stage in
a) PROF_DIR=${rnd}f_c3 bash tools/collect_profiles.sh && python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err && tail -c 300 gpurun_out/bench_default.json ;;
b) BENCH_ARGS="--config c2" PROF_STEPS=200 PLAIN_STEPS=200 PROF_DIR=${rnd}f_c2 bash tools/collect_profiles.sh &&
   BENCH_ARGS="--config c1" PROF_STEPS=200 PLAIN_STEPS=200 PROF_DIR=${rnd}f_c1 bash tools/collect_profiles.sh &&
   BENCH_ARGS="--config mansion_lr" PROF_DIR=${rnd}f_mansion bash tools/collect_profiles.sh &&
   BENCH_ARGS="--config mansion_151" PROF_DIR=${rnd}f_mansion151 bash tools/collect_profiles.sh ;;
c) BENCH_ARGS="--config c5 --rows 16" PROF_DIR=${rnd}f_c5s16 bash tools/collect_profiles.sh &&
   BENCH_ARGS="--config c5" PROF_STEPS=2 PLAIN_STEPS=3 PASS_TIMEOUT=400 PASSES="stats pmc_fetch pmc_write pmc_sq" PROF_DIR=${rnd}f_c5 bash tools/collect_profiles.sh ;;
c2) timeout -k 10 600 python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5_e2e.json 2> gpurun_out/bench_c5_e2e.err; tail -c 400 gpurun_out/bench_c5_e2e.json ;;
d) bash tools/profile_paths.sh && bash tools/context_lines.sh ;;
e) python3 tools/fuzz_parity.py 14000 4401 > gpurun_out/fuzz_parity_a.txt 2>&1; tail -1 gpurun_out/fuzz_parity_a.txt
   python3 tools/fuzz_sweep.py 2500 4402 > gpurun_out/fuzz_sweep_a.txt 2>&1; tail -1 gpurun_out/fuzz_sweep_a.txt
   FUZZ_ONLY_CHIP=1 python3 tools/fuzz_parity.py 600 4403 > gpurun_out/fuzz_chip_a.txt 2>&1; tail -1 gpurun_out/fuzz_chip_a.txt ;;
esac
