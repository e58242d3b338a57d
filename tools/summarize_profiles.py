#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (tools/collect_profiles.sh) into the committed summaries under profiles/.

    python tools/summarize_profiles.py r01 c3_n1

Writes profiles/<round>_<tag>_kernel_stats.csv (rocprofv3 --stats, ours kernels only),
profiles/<round>_<tag>_pmc.json (per-launch counter means for the rslf kernels) and updates
profiles/k2_traffic.json, which bench.py reads for roofline.traffic.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of a coalesced streaming read, so the read side is
calibrated on k0_pack of the same run, whose read byte count is known exactly
(V*S*U*C*4 of dense input): factor = known / reported, applied to the scan kernel's FETCH_SIZE.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "gpurun_out", os.environ.get("PROF_DIR", "prof"))


def pmc_means(sub):
    files = sorted(glob.glob(os.path.join(PROF, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not files:
        return {}
    rows = list(csv.DictReader(open(files[-1])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        if "rslf::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: dict(launches=len(disp[k]), **{c: v / len(disp[k]) for c, v in cs.items()}) for k, cs in agg.items()}


def main():
    rnd, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    # kernel stats
    st = sorted(glob.glob(os.path.join(PROF, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if st:
        rows = list(csv.reader(open(st[-1])))
        keep = [rows[0]] + [r for r in rows[1:] if "rslf::" in r[0]]
        with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)), "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_ALL).writerows(keep)
    summary = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_l2", "pmc_ta", "pmc_td", "pmc_tcp"):
        for k, v in pmc_means(sub).items():
            summary.setdefault(k, {}).update(v)
    bench = {}
    bj = os.path.join(PROF, "bench_stats.json")
    if os.path.exists(bj):
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            bench = {}
    # HBM traffic of the scan kernel, read side calibrated on k0_pack
    k2 = next((k for k in summary if "k2_scan" in k), None)
    k0 = next((k for k in summary if "k0_pack" in k), None)
    traffic = None
    if k2 and "FETCH_SIZE" in summary[k2]:
        factor, known = 2.0, None
        wl = bench.get("config", {}).get("workload", "")
        if k0 and "FETCH_SIZE" in summary[k0] and bench:
            import re
            m = re.search(r"(\d+)x(\d+) px x (\d+) views x (\d+) ch", wl)
            if m:
                U, V, S, C = map(int, m.groups())
                known = V * S * U * C * 4
                factor = known / (summary[k0]["FETCH_SIZE"] * 1024.0)
        rd = summary[k2]["FETCH_SIZE"] * 1024.0 * factor
        wr = summary[k2].get("WRITE_SIZE", 0.0) * 1024.0
        from remotesensingproject_amd import build as hb
        import datetime
        traffic = dict(source_hash=hb.source_hash(), collected=datetime.date.today().isoformat(),
                       hbm_bytes_per_launch=rd + wr, read_bytes=rd, write_bytes=wr,
                       fetch_size_kib=summary[k2]["FETCH_SIZE"], write_size_kib=summary[k2].get("WRITE_SIZE"),
                       read_calibration_factor=factor, calibrated_on="k0_pack known read bytes %s" % known,
                       kernel=k2, workload=wl)
        tpath = os.path.join(out, "k2_traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj[tag] = traffic
        json.dump(tj, open(tpath, "w"), indent=1)
    # effective shader clock of the scan under profiling: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)
    clock = None
    stats_avg_ns = None
    if st:
        for r in rows[1:]:
            if k2 and r[0].split("(")[0].replace("void ", "") == k2:
                stats_avg_ns = float(r[3]) if len(r) > 3 else None
    if k2 and "GRBM_GUI_ACTIVE" in summary.get(k2, {}) and stats_avg_ns:
        clock = summary[k2]["GRBM_GUI_ACTIVE"] / 8.0 / stats_avg_ns   # cycles per ns = GHz
    # several launches make one step of a grouped streaming / on-chip sweep: what bench.py's events bracket is their SUM
    launches_per_step = None
    if k2 and st:
        for r in rows[1:]:
            if r[0].split("(")[0].replace("void ", "") == k2 and bench.get("steps"):
                launches_per_step = float(r[1]) / (bench["steps"] + bench.get("warmup", 0) + 1)   # + the stats run
    if traffic and launches_per_step and launches_per_step > 1.5:
        n = round(launches_per_step)
        traffic.update(launches_per_step=n, per_launch_mean=dict(read_bytes=traffic["read_bytes"], write_bytes=traffic["write_bytes"]),
                       hbm_bytes_per_launch=(rd + wr) * n, read_bytes=rd * n, write_bytes=wr * n,
                       note="a step of this workload is %d launches of the kernel (row blocks); the figures are their SUM, what bench.py's HIP events bracket" % n)
        tj = json.load(open(tpath))
        tj[tag] = traffic
        json.dump(tj, open(tpath, "w"), indent=1)
    # the un-profiled line of the same command on the same lease (tools/collect_profiles.sh): the roofline of record
    plain = {}
    pj = os.path.join(PROF, "bench_plain.json")
    if os.path.exists(pj):
        try:
            plain = json.loads([l for l in open(pj).read().strip().splitlines() if l.startswith("{")][-1])
            json.dump(plain, open(os.path.join(out, "%s_bench_%s.json" % (rnd, tag)), "w"), indent=1)
        except Exception:  # noqa: BLE001
            plain = {}
    check = None
    if plain and stats_avg_ns and k2:
        rl = plain["roofline"]
        n = traffic.get("launches_per_step", 1) if traffic else 1
        frac_csv = rl["units_per_launch"] * rl["flops_per_unit"] / (stats_avg_ns * n * 1e-9) / 1e12 / rl["peak"]
        check = dict(frac_of_the_unprofiled_line=rl["frac"], frac_from_the_rocprof_average=frac_csv, ratio=frac_csv / rl["frac"],
                     kernel_ms_hip_events=rl["kernel_ms"], kernel_ms_rocprof_average=stats_avg_ns * n / 1e6,
                     ms_per_step_unprofiled=plain["ms_per_step"],
                     note="units x flops / rocprof average / peak against the un-profiled line's frac: same lease, same library")
    json.dump(dict(bench_line_under_rocprof=bench, bench_line_unprofiled_same_lease=plain, roofline_check=check,
                   per_launch_counter_means=summary, k2_traffic=traffic,
                   effective_clock_ghz_under_profiling=clock, kernel_avg_ns_rocprof_stats=stats_avg_ns,
                   source_hash=traffic.get("source_hash") if traffic else None),
              open(os.path.join(out, "%s_%s_pmc.json" % (rnd, tag)), "w"), indent=1)
    print(json.dumps(dict(k2=summary.get(k2), traffic=traffic), indent=1))


if __name__ == "__main__":
    main()
