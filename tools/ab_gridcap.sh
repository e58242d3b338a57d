#!/bin/bash
# A/B of the streaming kernel's row-tile grid (one workgroup per tile vs a capped grid striding over the tiles): tools/ab_gridcap.sh lib...
R=${GRAFT_REPO_ROOT:-$(pwd)}
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {}); print('$1', '%.4f ms/step' % j['ms_per_step'], 'K2 %.4f ms' % r.get('kernel_ms', float('nan')), 'frac %.4f' % r.get('frac', float('nan')))"; }
for rnd in 1 2; do
  for l in "$@"; do
    L=$(readlink -f $l); n=$(basename $l .so)
    RSLF_LIBRARY=$L python3 $R/bench.py --config mansion_lr --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n mansion dense"
    RSLF_LIBRARY=$L python3 $R/bench.py --path f2c --config mansion_lr --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | line "$n mansion f2c  "
    if [ -n "$REG_TOO" ]; then
      RSLF_LIBRARY=$L python3 $R/bench.py --path f2c --config skysat_lr --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "$n skysat f2c   "
      RSLF_LIBRARY=$L python3 $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c3           "
      RSLF_LIBRARY=$L python3 $R/bench.py --config c1 --steps 200 --warmup 40 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c1           "
    fi
  done
done
