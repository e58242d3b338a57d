#!/bin/bash
# Developer probe: a few counters of the scan kernel for one quick_bench shape and hook set.
#   tools/pmc_quick.sh TAG "U V S C D dmin dmax" "COUNTERS..." [ENV=VAL ...]   -> gpurun_out/pmc_quick/TAG.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shape=$2; counters=$3; shift 3
for e in "$@"; do export $e; done
OUT=$R/gpurun_out/pmc_quick/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT -- python3 $R/tools/quick_bench.py $shape > $OUT/log.txt 2>&1 || { echo "$tag failed"; tail -3 $OUT/log.txt; exit 1; }
python3 - <<PY | tee $R/gpurun_out/pmc_quick/$tag.txt
import csv, glob, collections
f = sorted(glob.glob("$OUT/*/*counter_collection.csv"))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k2_scan" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("$tag %-40s %-28s %.6g (mean of %d launches)" % (k[-40:], c, sum(v) / len(v), len(v)))
print(open("$OUT/log.txt").read().strip().splitlines()[-1])
PY
