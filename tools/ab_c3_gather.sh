#!/bin/bash
# c3's gather batch on evidence.  Same box, alternating runs of the shipped library (A) and a variant
# (B), N alternations, then HBM FETCH / WRITE counters of both (separate --pmc passes).  tools/ab_c3_gather.sh libA.so libB.so [N]
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$(readlink -f $1); B=$(readlink -f $2); N=${3:-4}
OUT=$R/gpurun_out/ab_c3
mkdir -p $OUT
: > $OUT/ab.txt
for rnd in $(seq 1 $N); do
  for l in $A $B; do
    RSLF_LIBRARY=$l python3 $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rnd', '$(basename $l)', 'K2 %.3f ms  step %.3f ms  frac %.4f' % (j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['frac']))" | tee -a $OUT/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for l in $A $B; do
  n=$(basename $l .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    RSLF_LIBRARY=$l timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${n}_$c -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/${n}_$c.log 2>&1 || exit 1
  done
  python3 - <<PY | tee -a $OUT/ab.txt
import csv, glob
def mean(c):
    f = sorted(glob.glob("$OUT/${n}_%s/*/*counter_collection.csv" % c))[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k2_scan" in r["Kernel_Name"] and r["Counter_Name"] == c]
    return sum(v) / len(v) * 1024.0
rd, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
alg = 889.6e6
print("$n: FETCH_SIZE %.0f KiB (x2 on gfx950 = %.0f MB read)  WRITE_SIZE %.0f MB  total %.0f MB = %.2fx algorithmic" % (rd / 1024, 2 * rd / 1e6, wr / 1e6, (2 * rd + wr) / 1e6, (2 * rd + wr) / alg))
PY
done
