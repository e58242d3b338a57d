#!/bin/bash
# A/B of two builds on the small configs at steady clocks (200 steps each, interleaved): tools/ab_small.sh libA.so libB.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in c2 c1; do
  for rnd in 1 2 3; do
    for l in "$1" "$2"; do
      RSLF_LIBRARY=$(readlink -f $l) python3 $R/bench.py --config $cfg --steps 200 --warmup 40 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$rnd', '$(basename $l)', 'K2 %.4f ms  step %.4f ms  frac %.4f' % (j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['frac']))"
    done
  done
done
