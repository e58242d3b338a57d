#!/bin/bash
# A/B of builds on the 2-D sweep (c3 and c2 volumes), with the per-kernel totals of one profiled run each:
#   tools/ab_sweep.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  for rnd in 1 2; do
    for cfg in c3 c2; do
      RSLF_LIBRARY=$(readlink -f $R/$l) python3 $R/bench.py --path sweep2d --config $cfg --steps $([ $cfg = c3 ] && echo 3 || echo 20) --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rnd', '$(basename $l)', '$cfg', '%.3f ms' % j['ms_per_step'])"
    done
  done
  D=$R/gpurun_out/ab_sweep_$(basename $l .so)
  rm -rf $D
  RSLF_LIBRARY=$(readlink -f $R/$l) rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --path sweep2d --config c3 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$D/*/*kernel_stats.csv"))[-1]
for r in list(csv.reader(open(f)))[1:5]:
    print("     %-40s calls %4s avg %8.1f us" % (r[0].replace("void rslf::", "")[:40], r[1], float(r[3]) / 1e3))
PY
done
