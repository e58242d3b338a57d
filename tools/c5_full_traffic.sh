#!/bin/bash
# HBM traffic of the streaming scan at FULL c5 size (two PMC passes: FETCH_SIZE, WRITE_SIZE) -> gpurun_out/prof_c5full
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c5full
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e"
echo start >> $OUT/progress.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
echo stats >> $OUT/progress.txt
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
echo fetch >> $OUT/progress.txt
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
echo write >> $OUT/progress.txt
