# developer A/B: kernel stats of a fine-to-coarse run for several builds (RSLF_LIBRARY) on one box
# usage: bash tools/ab_f2c.sh "<bench args>" lib1.so lib2.so ...
export TMPDIR=/tmp
ARGS="$1"; shift
for so in "$@"; do
  name=$(basename $so .so)
  RSLF_LIBRARY=$so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$name -o p -- python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || exit 1
  echo "== $name $(python3 -c "import json;print(json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step (under the profiler)" >> gpurun_out/ab_f2c.txt
  python3 - "$name" >> gpurun_out/ab_f2c.txt <<'PY'
import csv,sys
for r in csv.reader(open('gpurun_out/ab_%s/p_kernel_stats.csv' % sys.argv[1])):
    if r[0] != 'Name' and 'rslf' in r[0] and float(r[2]) > 2e6:
        print('  %-40s calls %5s total %9.2f ms avg %8.1f us' % (r[0].replace('void ','')[:40], r[1], float(r[2])/1e6, float(r[3])/1e3))
PY
done
cat gpurun_out/ab_f2c.txt
