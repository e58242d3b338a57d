#!/bin/bash
# The streaming kernel on the MansionLR shape under a few debug hooks (workgroups per tile, LDS share): no rebuild needed.
R=${GRAFT_REPO_ROOT:-$(pwd)}
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {}); print('$1', '%.4f ms/step' % j['ms_per_step'], 'K2 %.4f ms' % r.get('kernel_ms', float('nan')), 'frac %.4f' % r.get('frac', float('nan')))"; }
for h in "" "stream_groups=2" "stream_groups=4" "stream_groups=15" "stream_lds_kib=72" "stream_lds_kib=64"; do
  RSLF_BENCH_HOOKS=$h python3 $R/bench.py --config mansion_lr --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "[$h] dense"
  RSLF_BENCH_HOOKS=$h python3 $R/bench.py --path f2c --config mansion_lr --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | line "[$h] f2c  "
done
