#!/bin/bash
# A/B of the streaming kernel's RGB resident-prefix length: dense Mansion-like pile step (time + HBM traffic), its packed
# and pixel-per-wave forms, the c5 slice on the streaming kernel.   tools/ab_nres.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="1146 720 100 3 120 0 4"
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  L=$(readlink -f $R/$l); n=$(basename $l .so)
  echo "== $n"
  RSLF_LIBRARY=$L python3 $R/tools/quick_bench.py $S 2>&1 | tail -1
  RSLF_LIBRARY=$L PACKED=1 PX=0 python3 $R/tools/quick_bench.py $S 2>&1 | tail -1
  RSLF_LIBRARY=$L PACKED=1 PX=1 python3 $R/tools/quick_bench.py $S 2>&1 | tail -1
  RSLF_LIBRARY=$L FORCE_SCAN=2 python3 $R/tools/quick_bench.py 4096 16 201 3 512 -2 5.984375 2>&1 | tail -1
  for c in FETCH_SIZE WRITE_SIZE; do
    RSLF_LIBRARY=$L timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/ab_nres/${n}_$c -- python3 $R/tools/quick_bench.py $S > /dev/null 2>&1
    python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/ab_nres/${n}_$c/*/*counter_collection.csv"))[-1]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k2_scan" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
print("   $c %.0f KiB per launch%s" % (sum(v) / len(v), " (x2 on gfx950 = %.0f MB)" % (2 * sum(v) / len(v) * 1024 / 1e6) if "$c" == "FETCH_SIZE" else " (= %.0f MB)" % (sum(v) / len(v) * 1024 / 1e6)))
PY
  done
done
