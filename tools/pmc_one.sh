#!/bin/bash
# One PMC pass over bench.py: tools/pmc_one.sh <tag> <counters...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_one/$T -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/pmc_one/$T.json 2> $R/gpurun_out/pmc_one/$T.err
