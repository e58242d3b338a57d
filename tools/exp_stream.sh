# developer experiment: c5 slice (16 scanlines), streaming-kernel hooks; appends to gpurun_out/exp_stream.txt
# usage: bash tools/exp_stream.sh "stage=1,groups=16,lds=72" "stage=0,groups=16,lds=64" ...
for spec in "$@"; do
  echo "$spec" >> gpurun_out/exp_stream.txt
  RSLF_BENCH_HOOKS="$spec" timeout -k 10 200 python3 bench.py --config c5 --rows ${ROWS:-16} --steps 3 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   K2 %.3f ms  %.2f TFLOP/s' % (j['roofline']['kernel_ms'], j['roofline']['achieved']))" >> gpurun_out/exp_stream.txt || exit 1
done
cat gpurun_out/exp_stream.txt
