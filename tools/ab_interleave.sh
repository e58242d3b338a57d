#!/bin/bash
# A/B of the row-tile launches' XCD mapping (scanlines dealt in turn vs contiguous eighths): tools/ab_interleave.sh libA.so libB.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {}); print('$1', '%.3f ms/step' % j['ms_per_step'], 'K2 %.4f ms' % r.get('kernel_ms', float('nan')), 'frac %.4f' % r.get('frac', float('nan')))"; }
for rnd in 1 2 3; do
  for l in "$1" "$2"; do
    L=$(readlink -f $l); n=$(basename $l .so)
    RSLF_LIBRARY=$L python3 $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c3     "
    RSLF_LIBRARY=$L python3 $R/bench.py --config c2 --steps 200 --warmup 40 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c2     "
    RSLF_LIBRARY=$L python3 $R/bench.py --config c1 --steps 200 --warmup 40 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c1     "
    RSLF_LIBRARY=$L python3 $R/bench.py --config c5 --rows 16 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n c5 s16 "
    RSLF_LIBRARY=$L python3 $R/bench.py --config mansion_lr --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "$n mansion"
  done
done
for l in "$1" "$2"; do echo "$(basename $l) banded:"; RSLF_LIBRARY=$(readlink -f $l) BANDS=1 python3 $R/tools/probe_density.py mansion_lr 0.28 2>&1 | tail -1; done
