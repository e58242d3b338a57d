#!/bin/bash
# Developer probe: HBM read / write bytes per launch of the scan kernel that a shape takes (FETCH_SIZE x 2, WRITE_SIZE; separate passes).
#   tools/probe_kernel_traffic.sh "U V S C D dmin dmax" [ENV=VAL ...]      -> gpurun_out/kernel_traffic.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kernel_traffic
mkdir -p $OUT
shape=$1; shift
for e in "$@"; do export $e; done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/quick_bench.py $shape > $OUT/log_$c.txt 2>&1 || exit 1
done
python3 - <<PY >> $R/gpurun_out/kernel_traffic.txt
import csv, glob
def mean(c):
    f = sorted(glob.glob("$OUT/%s/*/*counter_collection.csv" % c))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k2_scan" in r["Kernel_Name"] and r["Counter_Name"] == c]
    v = [float(r["Counter_Value"]) for r in rows]
    return sum(v) / len(v) * 1024.0, rows[0]["Kernel_Name"].split("(")[0]
(rd, name), (wr, _) = mean("FETCH_SIZE"), mean("WRITE_SIZE")
U, V, S, C, D = [int(x) for x in "$shape".split()[:5]]
alg = U * V * S * C * 4.0
print("%-36s %s %s: read %.0f MB  write %.0f MB  = %.2fx the slab (%.0f MB)" % ("$shape", "$*", name, 2 * rd / 1e6, wr / 1e6, (2 * rd + wr) / alg, alg / 1e6))
print(open("$OUT/log_WRITE_SIZE.txt").read().strip().splitlines()[-1])
PY
tail -2 $R/gpurun_out/kernel_traffic.txt
