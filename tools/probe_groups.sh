# developer probe: dense launches of several heights / shapes (the automatic hypothesis groups of short grids)
for r in 0 139 274 540; do python bench.py --rows $r --steps 8 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 rows $r', round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],4))"; done
for c in c2 c1; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(j['roofline']['kernel_ms'],4), round(j['roofline']['frac'],4))"; done
python bench.py --path sweep2d --config c2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep c2', round(j['ms_per_step'],3))"
python bench.py --path f2c --config c2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f2c c2', round(j['ms_per_step'],3))"
python tools/probe_e2e.py 2>&1 | tail -2
