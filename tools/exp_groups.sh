# developer experiment: c5 slice (16 scanlines) under hypothesis groups; appends to gpurun_out/exp_groups.txt
for g in ${GROUPS_LIST:-1 8 16}; do
  echo "groups=$g" >> gpurun_out/exp_groups.txt
  RSLF_FORCE_GROUPS=$g timeout -k 10 200 python3 bench.py --config c5 --rows 16 --steps 3 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['roofline']['kernel_ms'], j['roofline']['achieved'])" >> gpurun_out/exp_groups.txt || exit 1
done
cat gpurun_out/exp_groups.txt
