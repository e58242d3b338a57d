#!/bin/bash
# Run on the GPU box from the repo root: kernel stats + one bench line for the 2-D sweep and fine-to-coarse rows (c2).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_paths
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in sweep2d f2c; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$P -- python3 $R/bench.py --path $P --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_$P.err
  python3 $R/bench.py --path $P --config c2 --steps 5 --warmup 2 > $OUT/bench_${P}_plain.json 2> $OUT/bench_${P}_plain.err
done
echo done
