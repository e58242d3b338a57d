#!/bin/bash
# A/B of builds on the rows around the path: tools/ab_paths.sh libA.so libB.so ...   (sweeps at c3 and c2, fine-to-coarse at c2 and a SkysatLR-like shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rnd in 1 2; do
  for l in "$@"; do
    for args in "--path sweep2d --config c3 --steps 3 --warmup 1" "--path sweep2d --config c2 --steps 20 --warmup 5" "--path f2c --config c2 --steps 10 --warmup 3" "--path f2c --shape 960,540,100,1,120,-1,4 --steps 2 --warmup 1"; do
      RSLF_LIBRARY=$(readlink -f $l) python3 $R/bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rnd', '$(basename $l)', '%-60s' % '$args', '%.3f ms' % j['ms_per_step'])"
    done
  done
done
