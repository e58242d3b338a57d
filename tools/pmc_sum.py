#!/usr/bin/env python3
"""Per-launch means of the counters of one kernel from tools/pmc_passes.sh output: tools/pmc_sum.py <dir> <kernel substring>"""
import collections, csv, glob, os, sys
d, key = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
out = {c: v / len(disp[c]) for c, v in agg.items()}
for c in sorted(out):
    print("%-28s %.6g" % (c, out[c]))
if "SQ_INSTS_VALU" in out and "GRBM_GUI_ACTIVE" in out:
    cyc = out["GRBM_GUI_ACTIVE"] / 8
    print("cycles %.4g; clk per VALU instr per SIMD %.3f" % (cyc, cyc * 1024 / out["SQ_INSTS_VALU"]))
if "SQ_WAVE_CYCLES" in out:
    w = out["SQ_WAVE_CYCLES"]
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM_RD"):
        if c in out:
            print("%s / WAVE_CYCLES = %.3f" % (c, out[c] / w))
