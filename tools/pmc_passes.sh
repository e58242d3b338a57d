#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass per argument (a quoted counter list), each under its own timeout and
# joined so that a pass that hangs ends the chain.  Output: gpurun_out/$PROF_DIR/pmc_<n>/ + a progress file.
#   BENCH_ARGS="--config c5 --rows 16" PROF_DIR=prof_c5b bash tools/pmc_passes.sh "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof_x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
n=0
for counters in "$@"; do
  n=$((n+1))
  echo "pass $n: $counters" >> $OUT/progress.txt
  timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py $ARGS > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  rc=$?
  echo "pass $n rc=$rc" >> $OUT/progress.txt
  if [ $rc -ne 0 ]; then echo "pass $n failed (rc $rc): stopping"; tail -3 $OUT/bench_$n.err; exit 1; fi
done
echo all passes done
