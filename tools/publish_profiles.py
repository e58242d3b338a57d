#!/usr/bin/env python3
"""Copy what tools/collect_profiles.sh / profile_paths.sh / context_lines.sh left under gpurun_out/ into profiles/ (round tag
as argv[1], e.g. r03): summaries of the K2 configs that were collected, the rows around the path, the context lines and
the default / c5 bench lines -- whatever is present -- then print the lines of record."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
def lastline(f):
    return [l for l in open(f) if l.startswith("{")][-1]
def stats(src_glob, dst):
    f = sorted(glob.glob(src_glob), key=os.path.getmtime)[-1]   # (the newest: the directories keep every earlier run)
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "rslf::" in r[0]]
    csv.writer(open(dst, "w", newline=""), quoting=csv.QUOTE_ALL).writerows(keep)
TAGS = {rnd + "f_c3": "c3_n1", rnd + "f_c2": "c2_n1", rnd + "f_c1": "c1_n1", rnd + "f_c5s16": "c5_slice16", rnd + "f_c5": "c5_n1",
        rnd + "f_mansion": "mansion_lr_n1",   # (the dense pile step of the report's MansionLR shape, 100 views RGB)
        rnd + "f_mansion151": "mansion_151_n1"}   # (the same frame with 151 views: a rung of the on-chip ladder other than c5's)
for d, t in TAGS.items():
    if os.path.isdir(os.path.join("gpurun_out", d)):
        subprocess.run([sys.executable, "tools/summarize_profiles.py", rnd, t], env=dict(os.environ, PROF_DIR=d), stdout=subprocess.DEVNULL, check=True)
for p in ("sweep2d", "f2c"):
    f = "gpurun_out/prof_paths/bench_%s_plain.json" % p
    if os.path.exists(f):
        open("profiles/%s_bench_%s_c2.json" % (rnd, p), "w").write(lastline(f))
        stats("gpurun_out/prof_paths/%s/*/*kernel_stats.csv" % p, "profiles/%s_%s_c2_kernel_stats.csv" % (rnd, p))
for t in ("f2c_skysat_lr", "f2c_mansion_lr", "sweep2d_c3"):
    f = "gpurun_out/context/%s.json" % t
    if os.path.exists(f):
        open("profiles/%s_context_%s.json" % (rnd, t), "w").write(lastline(f))
        stats("gpurun_out/context/prof_%s/*/*kernel_stats.csv" % t, "profiles/%s_context_%s_kernel_stats.csv" % (rnd, t))
for src, dst in (("gpurun_out/bench_default.json", "bench_default"), ("gpurun_out/bench_c5_e2e.json", "bench_c5_e2e")):
    if os.path.exists(src):
        open("profiles/%s_%s.json" % (rnd, dst), "w").write(lastline(src))
k = json.load(open("profiles/k2_traffic.json"))
for t in TAGS.values():
    if not os.path.exists("profiles/%s_%s_pmc.json" % (rnd, t)):
        continue
    j = json.load(open("profiles/%s_%s_pmc.json" % (rnd, t))); rc = j["roofline_check"]; b = json.load(open("profiles/%s_bench_%s.json" % (rnd, t))); e = k[t]
    print("%-11s K2 %.3f ms frac %.4f csv %.4f ratio %.3f | step %.3f ms value %.0f | read %.3f write %.3f total %.3f GB clock %.2f hash %s" % (
        t, rc["kernel_ms_hip_events"], rc["frac_of_the_unprofiled_line"], rc["frac_from_the_rocprof_average"], rc["ratio"], b["ms_per_step"], b["value"],
        e["read_bytes"] / 1e9, e["write_bytes"] / 1e9, e["hbm_bytes_per_launch"] / 1e9, j.get("effective_clock_ghz_under_profiling"), j["source_hash"]))
for f in ("bench_sweep2d_c2", "bench_f2c_c2", "context_f2c_skysat_lr", "context_f2c_mansion_lr", "context_sweep2d_c3"):
    if not os.path.exists("profiles/%s_%s.json" % (rnd, f)):
        continue
    j = json.load(open("profiles/%s_%s.json" % (rnd, f))); r = j.get("roofline", {})
    print("%-28s %6d M units/s  %.3f ms  frac %.4f k2_share %.3f  cpu %.1f" % (f, round(j["value"]), j["ms_per_step"], r.get("frac", float("nan")),
                                                                            r.get("k2_share", float("nan")), j.get("cpu_baseline", {}).get("value", float("nan"))))
j = json.load(open("profiles/%s_bench_default.json" % rnd)); print("default: value %.0f step %.2f ms frac %.4f e2e %.1f ms cpu %.1f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["e2e"]["ms"], j["cpu_baseline"]["value"]))
j = json.load(open("profiles/%s_bench_c5_e2e.json" % rnd)); print("c5:      value %.0f step %.1f ms frac %.4f e2e %.1f ms" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["e2e"]["ms"]))
