#!/bin/bash
# Developer probe: speed and HBM traffic of the on-chip kernel against the number of workgroups that share a tile
# (c5 slice of $ROWS scanlines; FETCH_SIZE / WRITE_SIZE in separate passes).  -> gpurun_out/chip_groups.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/chip_groups
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROWS=${ROWS:-64}
for g in ${GROUPS_LIST:-16 4 2 1}; do
  ARGS="--config c5 --rows $ROWS --steps 2 --warmup 1 --no-cpu-baseline --no-e2e"
  RSLF_BENCH_HOOKS=groups=$g timeout -k 5 200 python3 $R/bench.py $ARGS 2>/dev/null > $OUT/plain_$g.json || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    RSLF_BENCH_HOOKS=groups=$g timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_$g -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || exit 1
  done
  python3 - <<PY >> $R/gpurun_out/chip_groups.txt
import csv, glob, json
j = json.loads([l for l in open("$OUT/plain_$g.json") if l.startswith("{")][-1])
def mean(c):
    f = sorted(glob.glob("$OUT/%s_$g/*/*counter_collection.csv" % c))[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k2_scan" in r["Kernel_Name"] and r["Counter_Name"] == c]
    return sum(v) / len(v) * 1024.0
rd, wr = 2.0 * mean("FETCH_SIZE"), mean("WRITE_SIZE")
alg = j["roofline_hbm"]["achieved"] * j["roofline"]["kernel_ms"] * 1e6
print("groups %2d rows $ROWS: K2 %.2f ms  frac %.4f  read %.0f MB  write %.0f MB  total %.2fx algorithmic (%.0f MB)" % ($g, j["roofline"]["kernel_ms"], j["roofline"]["frac"], rd / 1e6, wr / 1e6, (rd + wr) / alg, alg / 1e6))
PY
done
cat $R/gpurun_out/chip_groups.txt
