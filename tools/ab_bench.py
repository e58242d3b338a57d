"""A/B timing of two builds of librslf_hip.so on the same GPU, interleaved rounds, one process each
(ctypes cannot unload a library), same volume.   python tools/ab_bench.py libA.so libB.so [config] [rows]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
rows = sys.argv[4] if len(sys.argv) > 4 else "0"
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, RSLF_LIBRARY=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--rows", rows, "--steps", "5", "--warmup", "2",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        res[l].append(j["roofline"]["kernel_ms"])
        print(rnd, os.path.basename(l), "K2 %.3f ms  step %.3f ms" % (j["roofline"]["kernel_ms"], j["ms_per_step"]), flush=True)
for l in libs:
    v = sorted(res[l])
    print(os.path.basename(l), "median %.3f min %.3f" % (v[len(v)//2], v[0]))
