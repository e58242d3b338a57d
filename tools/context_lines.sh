#!/bin/bash
# Published-shape context lines (report/rs_report.tex:427-437 shapes, synthetic fields) + the 2-D sweep at c3 size.
# Writes gpurun_out/context/*.json and rocprofv3 kernel stats; summarised into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/context
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, args...   (ONLY=tag runs just that one)
  tag=$1; shift
  if [ -n "$ONLY" ] && [ "$ONLY" != "$tag" ]; then return 0; fi
  echo "== $tag" >> $OUT/progress.txt
  timeout -k 10 500 python3 $R/bench.py "$@" --steps 2 --warmup 1 > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; return 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof_$tag.err || { echo "$tag profile failed"; return 1; }
  echo "done $tag" >> $OUT/progress.txt
}
run f2c_skysat_lr --path f2c --config skysat_lr &&
run f2c_mansion_lr --path f2c --config mansion_lr &&
run sweep2d_c3 --path sweep2d --config c3
cat $OUT/progress.txt
