#!/bin/bash
# Developer probe: which term of the on-chip kernel's HBM writes scales with what.  WRITE_SIZE of k2_scan_chip on synthetic
# fields that vary one dimension at a time (quick_bench.py U V S C D dmin dmax).  -> gpurun_out/chip_writes.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/chip_writes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while IFS="|" read -r shape envs; do
  [ -z "$shape" ] && continue
  export STREAM_SHARE= FORCE_GROUPS=
  [ -n "$envs" ] && export $envs
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$i -- python3 $R/tools/quick_bench.py $shape > $OUT/log_$i.txt 2>&1 || exit 1
  python3 - <<PY >> $R/gpurun_out/chip_writes.txt
import csv, glob
f = sorted(glob.glob("$OUT/pmc_$i/*/*counter_collection.csv"))[-1]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k2_scan_chip" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
U, V, S, C, D = [int(x) for x in "$shape".split()[:5]]
w = sum(v) / max(len(v), 1) * 1024.0
print("%-40s $envs launches %d  write %.1f MB  = %.3f B per (pixel, hypothesis), %.1f KB per 64-pixel tile" % ("$shape", len(v), w / 1e6, w / (U * V * D), w / (V * ((U + 62) // 63)) / 1e3))
PY
done <<SHAPES
4096 64 201 3 512 -2.0 5.96875|
4096 64 201 3 512 -2.0 5.96875|STREAM_SHARE=0
4096 64 201 3 512 -0.25 0.25|
4096 64 201 3 512 -0.25 0.25|STREAM_SHARE=0
4096 64 201 3 512 -0.25 0.25|FORCE_GROUPS=4
4096 64 201 3 512 -2.0 5.96875|FORCE_GROUPS=4
SHAPES
cat $R/gpurun_out/chip_writes.txt
