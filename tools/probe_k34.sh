# developer experiment: kernel stats of the c2 sweep for several builds (RSLF_LIBRARY); prints the k34/apply/packed-scan averages
export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so)
  RSLF_LIBRARY=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe_$name -o p -- python3 bench.py --path sweep2d --config c2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/probe_$name.log 2>&1 || exit 1
  echo "== $name" >> gpurun_out/probe.txt
  python3 - "$name" >> gpurun_out/probe.txt <<'PY'
import csv,sys
for r in csv.reader(open('gpurun_out/probe_%s/p_kernel_stats.csv' % sys.argv[1])):
    if r[0] != 'Name' and ('k34' in r[0] or 'apply' in r[0] or 'packed' in r[0]):
        print('  %-34s avg %7.1f us  min %7.1f  max %7.1f' % (r[0][:34], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))
PY
done
cat gpurun_out/probe.txt
